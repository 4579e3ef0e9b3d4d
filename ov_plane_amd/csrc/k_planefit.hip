// Plane fitting on the device: the step right in front of the plane update.
//   PlaneFitting::fit_plane / plane_fitting   track_plane/PlaneFitting.cpp:42-199   -> k_plane_ransac
//   PlaneFitting::optimize_plane              track_plane/PlaneFitting.cpp:201-514  -> k_plane_refine
// Batched over planes, one workgroup per plane.
//
// RANSAC.  The hypothesis sets come from std::shuffle with ONE std::mt19937(8888) per call whose state runs through all 200
// iterations (:93,:110) - inherently sequential, so the shuffles and the greedy 5-point selection (:113-135) are done while the
// batch is packed on the host (restated libstdc++ algorithms, see ovp_shuffle below), and the device scores the 200
// hypotheses in parallel: one lane per hypothesis, points and index sets staged in LDS.
//
// Refinement.  The reference hands the problem to Ceres (DENSE_SCHUR, DOGLEG, CauchyLoss(1), <= 12 iterations).  Here one
// thread owns one feature: it walks the feature's observations in order, keeps the feature's 3x3 blocks of J^T J (own block and
// the coupling to cp) and its part of the step in registers; the only cross-thread traffic is a handful of block-wide sums per
// iteration (deterministic: fixed butterfly inside a wave, waves added in order).  The trust-region loop is Ceres' own
// (trust_region_minimizer.cc / dogleg_strategy.cc with default options), in the arrowhead form the Schur complement gives it.
// Floating-point contraction is off so the iteration takes the decisions the scalar restatement takes.
#pragma clang fp contract(off)
#include "ovplane_hip.h"
#include "ovp_dev.h"
#include "ovp_kernels.h"

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

extern "C" int ovp_io_arena(ovp_ctx* c, size_t bytes, void** host, void** dev);  // ovp_api_ctx.hip: pinned host + device block per context

namespace ovp {

// ------------------------------------------------------------------------------------------------
// small dense helpers (wave-uniform or lane-local use)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void pf_sym3_eig(const double (&A)[9], double (&ev)[3]) {
  double a[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) a[i] = A[i];
  for (int sweep = 0; sweep < 30; ++sweep) {
    const double off = a[1] * a[1] + a[2] * a[2] + a[5] * a[5];
    // converged (same test as the oracle's; until round 6 all thirty sweeps ran for every RANSAC hypothesis)
    if (off <= 1e-34 * (a[0] * a[0] + a[4] * a[4] + a[8] * a[8])) break;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int q = p + 1; q < 3; ++q) {
        const double apq = a[3 * p + q];
        if (apq == 0.0) continue;
        const double theta = (a[3 * q + q] - a[3 * p + p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const double akp = a[3 * k + p], akq = a[3 * k + q];
          a[3 * k + p] = c * akp - s * akq;
          a[3 * k + q] = s * akp + c * akq;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const double apk = a[3 * p + k], aqk = a[3 * q + k];
          a[3 * p + k] = c * apk - s * aqk;
          a[3 * q + k] = s * apk + c * aqk;
        }
      }
  }
  ev[0] = a[0];
  ev[1] = a[4];
  ev[2] = a[8];
}

// symmetric positive definite 3x3 (lower part of S used): x = S^-1 b by Cholesky; false when a pivot is not positive
__device__ __forceinline__ bool pf_chol3_solve(const double (&S)[9], const double (&b)[3], double (&x)[3]) {
  const double d0 = S[0];
  if (!(d0 > 0.0)) return false;
  const double l00 = sqrt(d0);
  const double l10 = S[3] / l00, l20 = S[6] / l00;
  const double d1 = S[4] - l10 * l10;
  if (!(d1 > 0.0)) return false;
  const double l11 = sqrt(d1);
  const double l21 = (S[7] - l20 * l10) / l11;
  const double d2 = S[8] - l20 * l20 - l21 * l21;
  if (!(d2 > 0.0)) return false;
  const double l22 = sqrt(d2);
  const double y0 = b[0] / l00;
  const double y1 = (b[1] - l10 * y0) / l11;
  const double y2 = (b[2] - l20 * y0 - l21 * y1) / l22;
  x[2] = y2 / l22;
  x[1] = (y1 - l21 * x[2]) / l11;
  x[0] = (y0 - l10 * x[1] - l20 * x[2]) / l00;
  return isfinite(x[0]) && isfinite(x[1]) && isfinite(x[2]);
}

// deterministic block-wide sums of K values per thread: fixed xor butterfly inside the wave, wave partials added in order
template <int K>
__device__ __forceinline__ void pf_block_sum(double (&v)[K], double* red, int tid, int nwaves) {
#pragma unroll
  for (int k = 0; k < K; ++k) {
    double x = v[k];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) x += shfl_xor_f64(x, off);
    v[k] = x;
  }
  __syncthreads();  // red may still be read from the previous use
  if ((tid & 63) == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) red[(tid >> 6) * K + k] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; ++k) {
    double s = 0.0;
    for (int w = 0; w < nwaves; ++w) s += red[w * K + k];
    v[k] = s;
  }
}

// ------------------------------------------------------------------------------------------------
// fit of a plane to points through the normal equations:  A^T A x = -A^T 1,  abcd = [x, 1] / |x|
// (the reference solves the same least-squares problem with a column-pivoted QR, PlaneFitting.cpp:70-74)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool pf_plane_from_moments(const double (&M)[9], const double (&sv)[3], double cond_thresh,
                                                      bool cond_check, double (&abcd)[4]) {
  if (cond_check) {  // :59-66, sigma_i = sqrt(eig_i(A^T A))
    double ev[3];
    pf_sym3_eig(M, ev);
    const double lo = fmin(ev[0], fmin(ev[1], ev[2])), hi = fmax(ev[0], fmax(ev[1], ev[2]));
    const double cond = sqrt(hi > 0.0 ? hi : 0.0) / sqrt(lo > 0.0 ? lo : 0.0);
    if (cond > cond_thresh) return false;
  }
  double x[3];
  const double b[3] = {-sv[0], -sv[1], -sv[2]};
  if (!pf_chol3_solve(M, b, x)) return false;
  const double nn = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
  abcd[0] = x[0] / nn;
  abcd[1] = x[1] / nn;
  abcd[2] = x[2] / nn;
  abcd[3] = 1.0 / nn;
  const double c0 = -abcd[0] * abcd[3], c1 = -abcd[1] * abcd[3], c2 = -abcd[2] * abcd[3];
  return sqrt(c0 * c0 + c1 * c1 + c2 * c2) > 0.02;  // :77-80
}

struct RansacJob {
  const int* feat_start;   // [n_planes + 1]
  const double* pts;       // [F][3]
  const int* sets;         // [n_planes][200][5] local point indices, -1 = the call fails at this iteration (:138-141)
  int min_inlier_num;
  double max_cond;
  double* abcd;            // [n_planes][4]
  unsigned char* inlier;   // [F]
  unsigned char* ok;       // [n_planes]
};

static constexpr int RS_ITERS = 200;
static constexpr int RS_NMAX = 512;  // points of a plane staged in LDS (more: read from global memory)

__global__ __launch_bounds__(256) void k_plane_ransac(const RansacJob j) {
  const int pl = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int f0 = j.feat_start[pl], n = j.feat_start[pl + 1] - f0;
  const double* pts = j.pts + (size_t)3 * f0;
  const int* sets = j.sets + (size_t)pl * RS_ITERS * 5;
  __shared__ double h_abcd[RS_ITERS][4];
  __shared__ double h_err[RS_ITERS];
  __shared__ int h_cnt[RS_ITERS];  // -1 = hypothesis rejected by fit_plane, -2 = the call returns false here
  __shared__ int s_best;
  __shared__ double red[4 * 12];
  const double max_err = 0.05;
  const int ratio_n = (int)((double)n * 0.80);
  const int min_on_plane = j.min_inlier_num > ratio_n ? j.min_inlier_num : ratio_n;

  // One LANE per hypothesis (round 3).  The first version gave a hypothesis a wave (lanes over the points) and walked 50 of them per
  // wave, each behind two dependent global round trips (its index set, then the five points): 230 us per plane on average, 46 %
  // of the GPU time of a closed-loop frame with planes.  Points and index sets are staged in LDS once; the 3 x 3 fit and its
  // condition check are lane-local arithmetic anyway, and the score of a hypothesis is a walk over the points in index order
  // (every lane reads the same point: LDS broadcast) - the order in which the reference accumulates it (:147-155).
  __shared__ double pts_s[3 * RS_NMAX];
  __shared__ int sets_s[RS_ITERS * 5];
  const bool staged = n <= RS_NMAX;
  if (staged)
    for (int i = tid; i < 3 * n; i += 256) pts_s[i] = pts[i];
  for (int i = tid; i < RS_ITERS * 5; i += 256) sets_s[i] = sets[i];
  __syncthreads();
  const double* P = staged ? pts_s : pts;
  if (tid < RS_ITERS) {
    const int h = tid;
    const int* st = sets_s + 5 * h;
    int cnt = -1;
    double avg = 0.0, abcd[4] = {0.0, 0.0, 0.0, 0.0};
    if (st[0] < 0) {
      cnt = -2;
    } else {
      double M[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, sv[3] = {0, 0, 0};
      for (int s = 0; s < 5; ++s) {  // in set order
        const double* p = P + 3 * st[s];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          sv[a] += p[a];
#pragma unroll
          for (int b = 0; b < 3; ++b) M[3 * a + b] += p[a] * p[b];
        }
      }
      if (pf_plane_from_moments(M, sv, j.max_cond, true, abcd)) {  // :144
        int c = 0;
        double e = 0.0;
        for (int i = 0; i < n; ++i) {  // :147-155
          const double d = fabs(P[3 * i] * abcd[0] + P[3 * i + 1] * abcd[1] + P[3 * i + 2] * abcd[2] + abcd[3]);
          if (d < max_err) {
            ++c;
            e += d;
          }
        }
        cnt = c;
        avg = e / (double)c;
      }
    }
    h_cnt[h] = cnt;
    h_err[h] = avg;
#pragma unroll
    for (int k = 0; k < 4; ++k) h_abcd[h][k] = abcd[k];
  }
  __syncthreads();
  if (tid == 0) {  // the reference's sequential selection rule, :159-166
    int best = -1, best_cnt = 0;
    double best_err = -1.0;
    bool failed = (n < j.min_inlier_num);  // :97-100
    for (int h = 0; h < RS_ITERS && !failed; ++h) {
      const int c = h_cnt[h];
      if (c == -2) {
        failed = true;
        break;
      }
      if (c < 0) continue;
      const bool valid = (c > min_on_plane) && (h_err[h] < max_err);
      const bool better = (best_cnt < c) || (best_cnt == c && h_err[h] < best_err);
      if (valid && better) {
        best = h;
        best_cnt = c;
        best_err = h_err[h];
      }
    }
    s_best = failed ? -1 : best;
  }
  __syncthreads();
  const int best = s_best;
  bool ok = false;
  double abcd[4] = {0.0, 0.0, 0.0, 0.0};
  if (best >= 0) {  // :171-192: refit on the inliers of the best hypothesis, no condition check
    double v[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) v[k] = 0.0;
    const double a0 = h_abcd[best][0], a1 = h_abcd[best][1], a2 = h_abcd[best][2], a3 = h_abcd[best][3];
    for (int i = tid; i < n; i += 256) {
      const double* p = pts + 3 * i;
      const bool in = fabs(p[0] * a0 + p[1] * a1 + p[2] * a2 + a3) < max_err;
      j.inlier[f0 + i] = in ? 1 : 0;
      if (in) {
        v[0] += p[0] * p[0];
        v[1] += p[0] * p[1];
        v[2] += p[0] * p[2];
        v[3] += p[1] * p[1];
        v[4] += p[1] * p[2];
        v[5] += p[2] * p[2];
        v[6] += p[0];
        v[7] += p[1];
        v[8] += p[2];
      }
    }
    pf_block_sum<12>(v, red, tid, 4);
    const double M[9] = {v[0], v[1], v[2], v[1], v[3], v[4], v[2], v[4], v[5]};
    const double sv[3] = {v[6], v[7], v[8]};
    ok = pf_plane_from_moments(M, sv, j.max_cond, false, abcd);
  }
  if (!ok)
    for (int i = tid; i < n; i += 256) j.inlier[f0 + i] = 0;
  if (tid == 0) {
    j.ok[pl] = ok ? 1 : 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) j.abcd[4 * pl + k] = ok ? abcd[k] : 0.0;
  }
}

// ------------------------------------------------------------------------------------------------
// optimize_plane
// ------------------------------------------------------------------------------------------------
struct RefineJob {
  const int* feat_start;   // [n_planes + 1]
  const double* p0;        // [F][3]
  const int* obs_start;    // [F]
  const int* n_obs;        // [F]
  const double* uv;        // [O][2]
  const double* Rc;        // [O][9]
  const double* pc;        // [O][3]
  const double* cp0;       // [n_planes][3]
  const unsigned char* fix_plane;  // [n_planes]
  // the observations of a plane as one list, feature by feature in the features' order (built by the launcher): the evaluation of
  // the cost runs over this list with one thread per OBSERVATION, the feature's own thread then adds the pieces up in its order
  const int* item_start;   // [n_planes + 1]
  const int* item_ob;      // [items] observation index (into uv / Rc / pc)
  const int* item_lf;      // [items] feature of the observation, local index within its plane
  const int* feat_item0;   // [F] first item of the feature, relative to its plane's first
  double sigma_px_norm, sigma_c;
  double R_GtoC[9], p_CinG[3];     // current camera (from stateI, calib0: PlaneFitting.cpp:444-453)
  double* cp_out;          // [n_planes][3]
  double* p_out;           // [F][3]
  unsigned char* kept;     // [F]
  unsigned char* ok;       // [n_planes]
  int* iterations;         // [n_planes]
};

// what one feature contributes at the point (p, cp): cost, gradient and J^T J blocks (robust-loss corrected, unscaled)
struct FeatBlocks {
  double cost;
  double Hpp[6];  // xx xy xz yy yz zz
  double Hpc[9];  // row = p component, column = cp component
  double Hcc[6];
  double gp[3], gc[3];
};

// One point-on-plane residual block at (p, cp) (ceres/Factor_PointOnPlane.cpp:41-71, CauchyLoss(1) corrector): every observation of a
// feature adds the SAME block (PlaneFitting.cpp:366-368), so the feature's thread forms it once per evaluation and adds it m times.
struct ConsBlock {
  double cost;
  double JpJp[6], JcJc[6], JpJc[9], gp[3], gc[3];
};
template <bool WITH_J>
__device__ __forceinline__ void pf_constraint_block(const double (&p)[3], const double (&cp)[3], double sigma, ConsBlock& o) {
  const double d = sqrt(cp[0] * cp[0] + cp[1] * cp[1] + cp[2] * cp[2]);
  const double nv[3] = {cp[0] / d, cp[1] / d, cp[2] / d};
  const double np = nv[0] * p[0] + nv[1] * p[1] + nv[2] * p[2];
  const double w = 1.0 / sigma;
  const double r = -1.0 * w * (0.0 - (np - d));
  const double s = r * r;
  o.cost = 0.5 * log(1.0 + s);
  if (WITH_J) {
    const double a = sqrt(1.0 / (1.0 + s));
    const double ra = a * r;
    double Jp[3], Jc[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      Jp[k] = a * (w * nv[k]);
      Jc[k] = a * (w * 1.0 / d * (p[k] - np * nv[k] - d * nv[k]));
    }
    o.JpJp[0] = Jp[0] * Jp[0];
    o.JpJp[1] = Jp[0] * Jp[1];
    o.JpJp[2] = Jp[0] * Jp[2];
    o.JpJp[3] = Jp[1] * Jp[1];
    o.JpJp[4] = Jp[1] * Jp[2];
    o.JpJp[5] = Jp[2] * Jp[2];
    o.JcJc[0] = Jc[0] * Jc[0];
    o.JcJc[1] = Jc[0] * Jc[1];
    o.JcJc[2] = Jc[0] * Jc[2];
    o.JcJc[3] = Jc[1] * Jc[1];
    o.JcJc[4] = Jc[1] * Jc[2];
    o.JcJc[5] = Jc[2] * Jc[2];
#pragma unroll
    for (int a_ = 0; a_ < 3; ++a_) {
      o.gp[a_] = Jp[a_] * ra;
      o.gc[a_] = Jc[a_] * ra;
#pragma unroll
      for (int b_ = 0; b_ < 3; ++b_) o.JpJc[3 * a_ + b_] = Jp[a_] * Jc[b_];
    }
  }
}
template <bool WITH_J>
__device__ __forceinline__ void pf_add_constraint(FeatBlocks& o, const ConsBlock& c) {
  o.cost += c.cost;
  if (WITH_J) {
#pragma unroll
    for (int k = 0; k < 6; ++k) o.Hpp[k] += c.JpJp[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) o.Hcc[k] += c.JcJc[k];
#pragma unroll
    for (int a_ = 0; a_ < 3; ++a_) {
      o.gp[a_] += c.gp[a_];
      o.gc[a_] += c.gc[a_];
#pragma unroll
      for (int b_ = 0; b_ < 3; ++b_) o.Hpc[3 * a_ + b_] += c.JpJc[3 * a_ + b_];
    }
  }
}

// The reprojection block of ONE observation of a point p (PlaneFitting.cpp:330-364, CauchyLoss(1) corrector):
// out = [cost, Hpp (6), gp (3)].
static constexpr int PF_OB = 10;
template <bool WITH_J>
__device__ __forceinline__ void pf_reproj_block(const RefineJob& j, int ob, const double (&p)[3], double* out) {
  const double* R = j.Rc + (size_t)9 * ob;
  const double* c = j.pc + (size_t)3 * ob;
  const double dd[3] = {p[0] - c[0], p[1] - c[1], p[2] - c[2]};
  const double x = R[0] * dd[0] + R[1] * dd[1] + R[2] * dd[2];
  const double y = R[3] * dd[0] + R[4] * dd[1] + R[5] * dd[2];
  const double z = R[6] * dd[0] + R[7] * dd[1] + R[8] * dd[2];
  const double w = 1.0 / j.sigma_px_norm;
  const double r0 = w * (x / z - j.uv[2 * ob]), r1 = w * (y / z - j.uv[2 * ob + 1]);
  const double s = r0 * r0 + r1 * r1;
  out[0] = 0.5 * log(1.0 + s);
  if (WITH_J) {
    const double a = sqrt(1.0 / (1.0 + s));
    const double a0 = 1.0 / z, a2 = -x / (z * z), b2 = -y / (z * z);
    double J0[3], J1[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      J0[q] = a * (w * (a0 * R[q] + a2 * R[6 + q]));
      J1[q] = a * (w * (a0 * R[3 + q] + b2 * R[6 + q]));
    }
    const double ra0 = a * r0, ra1 = a * r1;
    out[1] = J0[0] * J0[0] + J1[0] * J1[0];
    out[2] = J0[0] * J0[1] + J1[0] * J1[1];
    out[3] = J0[0] * J0[2] + J1[0] * J1[2];
    out[4] = J0[1] * J0[1] + J1[1] * J1[1];
    out[5] = J0[1] * J0[2] + J1[1] * J1[2];
    out[6] = J0[2] * J0[2] + J1[2] * J1[2];
#pragma unroll
    for (int q = 0; q < 3; ++q) out[7 + q] = J0[q] * ra0 + J1[q] * ra1;
  }
}

// What every feature of the plane contributes at the points (p, cp): the threads of the workgroup take the plane's observations
// one each (a pass of blockDim.x observations at a time, their blocks staged in LDS), then the feature's own thread adds its
// observations' blocks and its constraint block in the order the sequential loop visits them (observation 0, constraint,
// observation 1, constraint, ...; PlaneFitting.cpp:330-368) - same sums, to the bit, as one thread walking its feature's list.
#ifdef OVP_PF_STAMPS
__device__ long long g_pf_stamps[64 * 12];
#define PF_T() __builtin_readcyclecounter()
#endif
struct EvalCtx {
  int tid, nthreads, nitems, it0;  // workgroup geometry; the plane's item range
  int my_item0, m;                 // this thread's feature: first item (relative), observations
  bool act, contrib;
  double* xs;   // [256][3] the features' points of this evaluation
  double* obs;  // [blockDim.x][PF_OB] one pass of observation blocks
};
template <bool WITH_J>
__device__ __forceinline__ void pf_eval_plane(const RefineJob& j, const EvalCtx& e, const double (&p)[3], const double (&cp)[3],
                                              FeatBlocks& o) {
  o.cost = 0.0;
  if (WITH_J) {
#pragma unroll
    for (int k = 0; k < 6; ++k) o.Hpp[k] = 0.0, o.Hcc[k] = 0.0;
#pragma unroll
    for (int k = 0; k < 9; ++k) o.Hpc[k] = 0.0;
#pragma unroll
    for (int k = 0; k < 3; ++k) o.gp[k] = 0.0, o.gc[k] = 0.0;
  }
  ConsBlock cb;
  if (e.contrib) {
    if (e.m == 0) {  // :276-279
      pf_constraint_block<WITH_J>(p, cp, 2.00 * j.sigma_c, cb);
      pf_add_constraint<WITH_J>(o, cb);
    } else {
      pf_constraint_block<WITH_J>(p, cp, j.sigma_c, cb);
    }
  }
  __syncthreads();  // (xs / obs may still be read from the previous evaluation)
  if (e.act) {
    e.xs[3 * e.tid] = p[0];
    e.xs[3 * e.tid + 1] = p[1];
    e.xs[3 * e.tid + 2] = p[2];
  }
  __syncthreads();
  for (int base = 0; base < e.nitems; base += e.nthreads) {
    const int i = base + e.tid;
    if (i < e.nitems) {
      const int ob = j.item_ob[e.it0 + i], lf = j.item_lf[e.it0 + i];
      const double pp[3] = {e.xs[3 * lf], e.xs[3 * lf + 1], e.xs[3 * lf + 2]};
      pf_reproj_block<WITH_J>(j, ob, pp, e.obs + (size_t)PF_OB * e.tid);
    }
    __syncthreads();
    if (e.contrib && e.m > 0) {
      const int lo = e.my_item0 > base ? e.my_item0 : base;
      const int hi = (e.my_item0 + e.m) < (base + e.nthreads) ? (e.my_item0 + e.m) : (base + e.nthreads);
      for (int i2 = lo; i2 < hi; ++i2) {
        const double* b = e.obs + (size_t)PF_OB * (i2 - base);
        o.cost += b[0];
        if (WITH_J) {
#pragma unroll
          for (int k = 0; k < 6; ++k) o.Hpp[k] += b[1 + k];
#pragma unroll
          for (int q = 0; q < 3; ++q) o.gp[q] += b[7 + q];
        }
        pf_add_constraint<WITH_J>(o, cb);
      }
    }
    __syncthreads();
  }
}

__device__ __forceinline__ double pf_sym_get(const double (&S)[6], int a, int b) {
  const int i = a < b ? a : b, k = a < b ? b : a;
  return S[i == 0 ? k : (i == 1 ? 2 + k : 5)];
}

__global__ __launch_bounds__(256) void k_plane_refine(const RefineJob j) {
  const int pl = blockIdx.x, tid = threadIdx.x, nwaves = (blockDim.x + 63) >> 6;
  const int f0 = j.feat_start[pl], nf = j.feat_start[pl + 1] - f0;
  const bool fix = j.fix_plane[pl] != 0;
  __shared__ double red[16 * 16];
  __shared__ double xs_s[256 * 3];
  __shared__ double obs_s[256 * PF_OB];
  const bool act = tid < nf;            // this thread owns feature f0 + tid
  const int f = f0 + (act ? tid : 0);
  const int m = act ? j.n_obs[f] : 0;
  const bool freef = act && m > 0;      // free parameter block
  const bool contrib = act && (freef || !fix);  // has residual blocks with a free parameter
#ifdef OVP_PF_STAMPS
  long long t_evalj = 0, t_evalc = 0, tp[6] = {0, 0, 0, 0, 0, 0};
  long long t_mark = 0;
  const long long t_begin = PF_T();
#endif
  EvalCtx ev;
  ev.tid = tid;
  ev.nthreads = blockDim.x;
  ev.it0 = j.item_start[pl];
  ev.nitems = j.item_start[pl + 1] - ev.it0;
  ev.my_item0 = act ? j.feat_item0[f] : 0;
  ev.m = m;
  ev.act = act;
  ev.contrib = contrib;
  ev.xs = xs_s;
  ev.obs = obs_s;
  double x[3], cp[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    x[k] = act ? j.p0[3 * (size_t)f + k] : 0.0;
    cp[k] = j.cp0[3 * pl + k];
  }
  const double p_old[3] = {x[0], x[1], x[2]};

  // ---- :214-217 ----
  bool early_fail = (!fix && nf < 4) || (fix && nf == 0);
  int nfree_tot;
  {
    double v[1] = {freef ? 1.0 : 0.0};
    pf_block_sum<1>(v, red, tid, nwaves);
    nfree_tot = (int)v[0];
  }
  const int npar = 3 * nfree_tot + (fix ? 0 : 3);
  bool converged = false;
  int iter = 0;
  if (!early_fail && npar == 0) converged = true;

  if (!early_fail && npar > 0) {
    const double function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;
    const double min_relative_decrease = 1e-3, min_radius = 1e-32;
    const double min_mu = 1e-8, max_mu = 1.0, mu_inc = 10.0;
    double radius = 1e4, mu = 1e-8;
    bool reuse = false;
    int invalid_run = 0;
    FeatBlocks B;
    double Hcc[6], gc[3];  // plane totals (unscaled)
    double cost, gmax, xnorm;
    double sp[3] = {1.0, 1.0, 1.0}, sc[3] = {1.0, 1.0, 1.0};  // Jacobi scaling, fixed at the first point
    // quantities of the current linearisation in the Jacobi-scaled space
    double Hs[6], Hsc[9], gs[3], dg[3];       // own block, coupling, gradient, diag
    double Hscc[6], gsc[3], dgc[3];           // cp
    double gn[3], gnc[3], alpha = 0.0, dogleg_norm = 0.0;

    auto linearise = [&](bool first) {
#ifdef OVP_PF_STAMPS
      const long long te0 = PF_T();
#endif
      pf_eval_plane<true>(j, ev, x, cp, B);
#ifdef OVP_PF_STAMPS
      t_evalj += PF_T() - te0;
#endif
      double v[13];
      v[0] = contrib ? B.cost : 0.0;
#pragma unroll
      for (int k = 0; k < 6; ++k) v[1 + k] = (contrib && !fix) ? B.Hcc[k] : 0.0;
#pragma unroll
      for (int k = 0; k < 3; ++k) v[7 + k] = (contrib && !fix) ? B.gc[k] : 0.0;
      // unscaled gradient max-norm and |x|^2 ride along (max through a sum is not possible: separate reduction below)
      v[10] = freef ? (x[0] * x[0] + x[1] * x[1] + x[2] * x[2]) : 0.0;
      v[11] = 0.0;
      v[12] = 0.0;
      pf_block_sum<13>(v, red, tid, nwaves);
      cost = v[0];
#pragma unroll
      for (int k = 0; k < 6; ++k) Hcc[k] = v[1 + k];
#pragma unroll
      for (int k = 0; k < 3; ++k) gc[k] = v[7 + k];
      xnorm = sqrt(v[10] + (fix ? 0.0 : cp[0] * cp[0] + cp[1] * cp[1] + cp[2] * cp[2]));
      // gradient max-norm
      double gm = 0.0;
      if (freef) gm = fmax(fabs(B.gp[0]), fmax(fabs(B.gp[1]), fabs(B.gp[2])));
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) gm = fmax(gm, shfl_xor_f64(gm, off));
      __syncthreads();
      if ((tid & 63) == 0) red[tid >> 6] = gm;
      __syncthreads();
      gm = 0.0;
      for (int w = 0; w < nwaves; ++w) gm = fmax(gm, red[w]);
      if (!fix) gm = fmax(gm, fmax(fabs(gc[0]), fmax(fabs(gc[1]), fabs(gc[2]))));
      gmax = gm;
      if (first) {  // Jacobi scaling 1 / (1 + |column|), estimated once (trust_region_minimizer.cc, Init)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          sp[k] = freef ? 1.0 / (1.0 + sqrt(pf_sym_get(B.Hpp, k, k))) : 1.0;
          sc[k] = 1.0 / (1.0 + sqrt(pf_sym_get(Hcc, k, k)));
        }
      }
    };
    linearise(true);

    while (!converged) {
      if (iter >= 12) break;  // :385 max_num_iterations -> NO_CONVERGENCE
      if (gmax <= gradient_tolerance || radius <= min_radius) {
        converged = true;
        break;
      }
      ++iter;
      bool step_ok = true;
#ifdef OVP_PF_STAMPS
      t_mark = PF_T();
#endif
      if (!reuse) {
        reuse = true;
        // scaled blocks, diag, gradient
        const int ia[6] = {0, 0, 0, 1, 1, 2}, ib[6] = {0, 1, 2, 1, 2, 2};
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          Hs[k] = freef ? B.Hpp[k] * sp[ia[k]] * sp[ib[k]] : 0.0;
          Hscc[k] = Hcc[k] * sc[ia[k]] * sc[ib[k]];
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          gs[a] = freef ? B.gp[a] * sp[a] : 0.0;
          gsc[a] = gc[a] * sc[a];
#pragma unroll
          for (int b = 0; b < 3; ++b) Hsc[3 * a + b] = (freef && !fix) ? B.Hpc[3 * a + b] * sp[a] * sc[b] : 0.0;
          double q = pf_sym_get(Hs, a, a);
          q = fmin(fmax(q, 1e-6), 1e32);
          dg[a] = sqrt(q);
          double qc = pf_sym_get(Hscc, a, a);
          qc = fmin(fmax(qc, 1e-6), 1e32);
          dgc[a] = sqrt(qc);
        }
#ifdef OVP_PF_STAMPS
        tp[0] += PF_T() - t_mark;
        t_mark = PF_T();
#endif
        // Cauchy point: alpha = |g/d|^2 / (v^T H v), v = g / d^2
        {
          double vv[3], vc[3];
#pragma unroll
          for (int a = 0; a < 3; ++a) {
            vv[a] = gs[a] / (dg[a] * dg[a]);
            vc[a] = fix ? 0.0 : gsc[a] / (dgc[a] * dgc[a]);
          }
          double r2[2] = {0.0, 0.0};
          if (freef) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
              r2[0] += (gs[a] / dg[a]) * (gs[a] / dg[a]);
#pragma unroll
              for (int b = 0; b < 3; ++b) r2[1] += vv[a] * pf_sym_get(Hs, a, b) * vv[b] + 2.0 * vv[a] * Hsc[3 * a + b] * vc[b];
            }
          }
          pf_block_sum<2>(r2, red, tid, nwaves);
          if (!fix) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
              r2[0] += (gsc[a] / dgc[a]) * (gsc[a] / dgc[a]);
#pragma unroll
              for (int b = 0; b < 3; ++b) r2[1] += vc[a] * pf_sym_get(Hscc, a, b) * vc[b];
            }
          }
          alpha = r2[0] / r2[1];
        }
#ifdef OVP_PF_STAMPS
        tp[1] += PF_T() - t_mark;
        t_mark = PF_T();
#endif
        // Gauss-Newton step: (H + mu D^2) y = g through the Schur complement on cp; mu grows on failure
        bool solved = false;
        while (mu < max_mu) {
#ifdef OVP_PF_STAMPS
          tp[4] += 1;
#endif
          double A[9], yf[3] = {0.0, 0.0, 0.0}, W[9];  // W = A^-1 Hsc (3x3), yf = A^-1 gs
          bool okf = true;
          if (freef) {
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
              for (int b = 0; b < 3; ++b) A[3 * a + b] = pf_sym_get(Hs, a, b) + (a == b ? mu * dg[a] * dg[a] : 0.0);
            okf = pf_chol3_solve(A, gs, yf);
            if (!fix) {
#pragma unroll
              for (int b = 0; b < 3; ++b) {
                const double col[3] = {Hsc[b], Hsc[3 + b], Hsc[6 + b]};
                double w3[3];
                okf = pf_chol3_solve(A, col, w3) && okf;
                W[b] = w3[0];
                W[3 + b] = w3[1];
                W[6 + b] = w3[2];
              }
            }
          }
          double v[10];
#pragma unroll
          for (int k = 0; k < 10; ++k) v[k] = 0.0;
          v[9] = okf ? 0.0 : 1.0;
          if (freef && !fix && okf) {
            // S -= Hsc^T W ; rhs -= Hsc^T yf
            const int ia2[6] = {0, 0, 0, 1, 1, 2}, ib2[6] = {0, 1, 2, 1, 2, 2};
#pragma unroll
            for (int k = 0; k < 6; ++k) {
              double s = 0.0;
#pragma unroll
              for (int a = 0; a < 3; ++a) s += Hsc[3 * a + ia2[k]] * W[3 * a + ib2[k]];
              v[k] = s;
            }
#pragma unroll
            for (int b = 0; b < 3; ++b) {
              double s = 0.0;
#pragma unroll
              for (int a = 0; a < 3; ++a) s += Hsc[3 * a + b] * yf[a];
              v[6 + b] = s;
            }
          }
          pf_block_sum<10>(v, red, tid, nwaves);
          bool okall = !(v[9] > 0.0);
          double yc[3] = {0.0, 0.0, 0.0};
          if (okall && !fix) {
            double S[9], rhs[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
              rhs[a] = gsc[a] - v[6 + a];
#pragma unroll
              for (int b = 0; b < 3; ++b) {
                const int i = a < b ? a : b, k = a < b ? b : a;
                const int idx = i == 0 ? k : (i == 1 ? 2 + k : 5);
                S[3 * a + b] = pf_sym_get(Hscc, a, b) + (a == b ? mu * dgc[a] * dgc[a] : 0.0) - v[idx];
              }
            }
            okall = pf_chol3_solve(S, rhs, yc);
          }
          if (okall) {
            // back substitution y_f = yf - W yc ; gn = -d * y
#pragma unroll
            for (int a = 0; a < 3; ++a) {
              double yy = yf[a];
              if (freef && !fix) yy -= W[3 * a] * yc[0] + W[3 * a + 1] * yc[1] + W[3 * a + 2] * yc[2];
              gn[a] = freef ? -dg[a] * yy : 0.0;
              gnc[a] = fix ? 0.0 : -dgc[a] * yc[a];
            }
            solved = true;
            break;
          }
          mu *= mu_inc;
        }
        if (!solved) step_ok = false;
#ifdef OVP_PF_STAMPS
        tp[2] += PF_T() - t_mark;
#endif
      }
#ifdef OVP_PF_STAMPS
      t_mark = PF_T();
#endif
      double st[3] = {0.0, 0.0, 0.0}, stc[3] = {0.0, 0.0, 0.0};
      double model_change = 0.0;
      if (step_ok) {
        // traditional dogleg on (g / d, gn)
        double r3[3] = {0.0, 0.0, 0.0};  // |gn|^2, |g/d|^2, (g/d).gn
        if (freef) {
#pragma unroll
          for (int a = 0; a < 3; ++a) {
            const double q = gs[a] / dg[a];
            r3[0] += gn[a] * gn[a];
            r3[1] += q * q;
            r3[2] += q * gn[a];
          }
        }
        pf_block_sum<3>(r3, red, tid, nwaves);
        if (!fix) {
#pragma unroll
          for (int a = 0; a < 3; ++a) {
            const double q = gsc[a] / dgc[a];
            r3[0] += gnc[a] * gnc[a];
            r3[1] += q * q;
            r3[2] += q * gnc[a];
          }
        }
        const double gn_norm = sqrt(r3[0]), gs_norm = sqrt(r3[1]);
        double cg, cn;  // step = cg * (g/d) + cn * gn
        bool need_norm = false;
        if (gn_norm <= radius) {
          cg = 0.0;
          cn = 1.0;
          dogleg_norm = gn_norm;
        } else if (gs_norm * alpha >= radius) {
          cg = -(radius / gs_norm);
          cn = 0.0;
          dogleg_norm = radius;
        } else {
          const double b_dot_a = -alpha * r3[2];
          const double a2 = alpha * alpha * gs_norm * gs_norm;
          const double bma2 = a2 - 2.0 * b_dot_a + gn_norm * gn_norm;
          const double c = b_dot_a - a2;
          const double dd = sqrt(c * c + bma2 * (radius * radius - a2));
          const double beta = (c <= 0) ? (dd - c) / bma2 : (radius * radius - a2) / (dd + c);
          cg = -alpha * (1.0 - beta);
          cn = beta;
          need_norm = true;
        }
        double r4[4] = {0.0, 0.0, 0.0, 0.0};  // |step_d|^2, step.g, step^T H step, |delta|^2
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          const double sd = freef ? cg * (gs[a] / dg[a]) + cn * gn[a] : 0.0;
          const double sdc = fix ? 0.0 : cg * (gsc[a] / dgc[a]) + cn * gnc[a];
          r4[0] += sd * sd;
          st[a] = freef ? sd / dg[a] : 0.0;
          stc[a] = fix ? 0.0 : sdc / dgc[a];
        }
        if (freef) {
#pragma unroll
          for (int a = 0; a < 3; ++a) {
            r4[1] += st[a] * gs[a];
            r4[3] += (st[a] * sp[a]) * (st[a] * sp[a]);
#pragma unroll
            for (int b = 0; b < 3; ++b) r4[2] += st[a] * pf_sym_get(Hs, a, b) * st[b] + 2.0 * st[a] * Hsc[3 * a + b] * stc[b];
          }
        }
        pf_block_sum<4>(r4, red, tid, nwaves);
        if (!fix) {
#pragma unroll
          for (int a = 0; a < 3; ++a) {
            const double sdc = cg * (gsc[a] / dgc[a]) + cn * gnc[a];
            r4[0] += sdc * sdc;
            r4[1] += stc[a] * gsc[a];
            r4[3] += (stc[a] * sc[a]) * (stc[a] * sc[a]);
#pragma unroll
            for (int b = 0; b < 3; ++b) r4[2] += stc[a] * pf_sym_get(Hscc, a, b) * stc[b];
          }
        }
        if (need_norm) dogleg_norm = sqrt(r4[0]);
        model_change = -(r4[1] + 0.5 * r4[2]);
#ifdef OVP_PF_STAMPS
        tp[3] += PF_T() - t_mark;
#endif
        if (!(model_change > 0.0)) step_ok = false;
        if (step_ok) {
          const double snorm = sqrt(r4[3]);
          double xc[3], cpc[3];
#pragma unroll
          for (int a = 0; a < 3; ++a) {
            xc[a] = x[a] + st[a] * sp[a];
            cpc[a] = cp[a] + stc[a] * sc[a];
          }
          FeatBlocks Bc;
#ifdef OVP_PF_STAMPS
          const long long tc0 = PF_T();
#endif
          pf_eval_plane<false>(j, ev, xc, cpc, Bc);
#ifdef OVP_PF_STAMPS
          t_evalc += PF_T() - tc0;
#endif
          double cv[1] = {contrib ? Bc.cost : 0.0};
          pf_block_sum<1>(cv, red, tid, nwaves);
          const double cand = cv[0];
          if (snorm <= parameter_tolerance * (xnorm + parameter_tolerance)) {
            converged = true;
            break;
          }
          if (fabs(cost - cand) <= function_tolerance * cost) {
            converged = true;
            break;
          }
          const double quality = (cost - cand) / model_change;
          if (quality > min_relative_decrease) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
              x[a] = xc[a];
              cp[a] = cpc[a];
            }
            linearise(false);
            if (quality < 0.25) radius *= 0.5;
            if (quality > 0.75) radius = fmax(radius, 3.0 * dogleg_norm);
            mu = fmax(min_mu, 2.0 * mu / mu_inc);
            reuse = false;
          } else {
            radius *= 0.5;
            reuse = true;
          }
          invalid_run = 0;
          continue;
        }
      }
      // HandleInvalidStep
      if (++invalid_run >= 5) break;
      mu *= mu_inc;
      reuse = false;
    }
  }

  // ---- :431-498 ----
  bool keep = false;
  if (converged && act) {
    const double d = sqrt(cp[0] * cp[0] + cp[1] * cp[1] + cp[2] * cp[2]);
    const double e = (p_old[0] * cp[0] + p_old[1] * cp[1] + p_old[2] * cp[2]) / d + (-d);  // old estimate, new plane (:467)
    const double dd[3] = {x[0] - j.p_CinG[0], x[1] - j.p_CinG[1], x[2] - j.p_CinG[2]};
    const double z = j.R_GtoC[6] * dd[0] + j.R_GtoC[7] * dd[1] + j.R_GtoC[8] * dd[2];
    const double nrm = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
    keep = (fabs(e) < 0.03) && !isnan(nrm) && !(z < 0.1);
  }
  double kv[1] = {keep ? 1.0 : 0.0};
  pf_block_sum<1>(kv, red, tid, nwaves);
  const int cnt = (int)kv[0];
  const int ratio_n = (int)((double)nf * 0.80);
  const int min_on_plane = 4 > ratio_n ? 4 : ratio_n;
  const bool fail = !converged || (nf != 1 && cnt < min_on_plane) || (fix && nf == 1 && cnt == 0);
  if (act) {
    const bool k = keep && !fail;
    j.kept[f] = k ? 1 : 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) j.p_out[3 * (size_t)f + a] = k ? x[a] : p_old[a];
  }
#ifdef OVP_PF_STAMPS
  if (tid == 0 && pl < 64) {
    long long* o = g_pf_stamps + 8 * pl;
    o[0] = t_evalj;
    o[1] = t_evalc;
    o[2] = PF_T() - t_begin;
    o[3] = iter;
    o[4] = nf;
    o[5] = ev.nitems;
    o[6] = tp[4];
    g_pf_stamps[64 * 8 + 4 * pl + 0] = tp[0];
    g_pf_stamps[64 * 8 + 4 * pl + 1] = tp[1];
    g_pf_stamps[64 * 8 + 4 * pl + 2] = tp[2];
    g_pf_stamps[64 * 8 + 4 * pl + 3] = tp[3];
  }
#endif
  if (tid == 0) {
    j.ok[pl] = fail ? 0 : 1;
    j.iterations[pl] = iter;
#pragma unroll
    for (int a = 0; a < 3; ++a) j.cp_out[3 * pl + a] = fail ? j.cp0[3 * pl + a] : cp[a];
  }
}

// ------------------------------------------------------------------------------------------------
// host side of the RANSAC: std::mt19937 + libstdc++'s std::shuffle / uniform_int_distribution, restated so that the
// hypothesis sets do not depend on the compiler this library is built with (variant 0 = GCC <= 10, the reference's
// platforms; 1 = GCC >= 11).
// ------------------------------------------------------------------------------------------------
struct Mt19937 {
  uint32_t mt[624];
  int idx;
  explicit Mt19937(uint32_t seed) {
    mt[0] = seed;
    for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    idx = 624;
  }
  uint32_t next() {
    if (idx >= 624) {
      for (int i = 0; i < 624; ++i) {
        const uint32_t y = (mt[i] & 0x80000000u) | (mt[(i + 1) % 624] & 0x7fffffffu);
        mt[i] = mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
      }
      idx = 0;
    }
    uint32_t y = mt[idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
  }
};

static uint64_t uniform_int(Mt19937& g, uint64_t urange, int variant) {
  const uint64_t urngrange = 0xFFFFFFFFull;
  if (urange >= urngrange) return g.next();
  const uint64_t uerange = urange + 1;
  if (variant == 0) {
    const uint64_t scaling = urngrange / uerange, past = uerange * scaling;
    uint64_t ret;
    do ret = g.next();
    while (ret >= past);
    return ret / scaling;
  }
  const uint32_t range = (uint32_t)uerange;
  uint64_t product = (uint64_t)g.next() * (uint64_t)range;
  uint32_t low = (uint32_t)product;
  if (low < range) {
    const uint32_t threshold = (uint32_t)(0u - range) % range;
    while (low < threshold) {
      product = (uint64_t)g.next() * (uint64_t)range;
      low = (uint32_t)product;
    }
  }
  return product >> 32;
}

static void ovp_shuffle(std::vector<int>& v, Mt19937& g, int variant) {
  const int n = (int)v.size();
  if (n <= 0) return;
  const uint64_t urngrange = 0xFFFFFFFFull, urange = (uint64_t)n;
  if (urngrange / urange >= urange) {
    int i = 1;
    if ((urange % 2) == 0) {
      std::swap(v[i], v[uniform_int(g, 1, variant)]);
      ++i;
    }
    while (i != n) {
      const uint64_t swap_range = (uint64_t)i + 1, b1 = swap_range + 1;
      const uint64_t xx = uniform_int(g, swap_range * b1 - 1, variant);
      std::swap(v[i], v[xx / b1]);
      ++i;
      std::swap(v[i], v[xx % b1]);
      ++i;
    }
    return;
  }
  for (int i = 1; i < n; ++i) std::swap(v[i], v[uniform_int(g, (uint64_t)i, variant)]);
}

}  // namespace ovp

#define PF_HIPCHK(x)                      \
  do {                                    \
    hipError_t e_ = (x);                  \
    if (e_ != hipSuccess) {               \
      if (blob) (void)hipFree(blob);      \
      return (int)e_;                     \
    }                                     \
  } while (0)

extern "C" int ovp_plane_fitting(ovp_ctx* c, const ovp_planefit_batch* b, double* abcd, uint8_t* inlier, uint8_t* ok) {
  if (!c || !b || !abcd || !inlier || !ok) return OVP_E_ARG;
  const int P = b->n_planes;
  if (P <= 0) return 0;
  const int F = b->feat_start[P];
  void* strm = nullptr;
  if (ovp_ctx_stream(c, &strm)) return OVP_E_ARG;
  hipStream_t s = (hipStream_t)strm;
  // hypothesis sets: PlaneFitting.cpp:104-141 for every plane
  std::vector<int> sets((size_t)P * ovp::RS_ITERS * 5, -1);
  for (int pl = 0; pl < P; ++pl) {
    const int f0 = b->feat_start[pl], n = b->feat_start[pl + 1] - f0;
    if (n < b->min_inlier_num) continue;  // the kernel reports the failure
    ovp::Mt19937 g(8888u);
    std::vector<int> order((size_t)n);
    bool dead = false;
    for (int it = 0; it < ovp::RS_ITERS && !dead; ++it) {
      for (int i = 0; i < n; ++i) order[i] = i;
      ovp::ovp_shuffle(order, g, b->shuffle_variant);
      int set[5], ns = 0;
      for (int k = 0; k < n && ns < 5; ++k) {
        const double* p = b->p_FinG + 3 * (size_t)(f0 + order[k]);
        bool good = true;
        for (int q = 0; q < ns; ++q) {
          const double* r = b->p_FinG + 3 * (size_t)(f0 + set[q]);
          const double d0 = r[0] - p[0], d1 = r[1] - p[1], d2 = r[2] - p[2];
          if (std::sqrt(d0 * d0 + d1 * d1 + d2 * d2) < 0.05) {
            good = false;
            break;
          }
        }
        if (ns == 0 || good) set[ns++] = order[k];
      }
      int* dst = sets.data() + ((size_t)pl * ovp::RS_ITERS + it) * 5;
      if (ns != 5) {
        dead = true;  // :138-141 - dst stays -1: the call fails at this iteration
      } else {
        for (int k = 0; k < 5; ++k) dst[k] = set[k];
      }
    }
  }
  // one pinned block in, one out (ovp_io_arena): [feat_start | points | hypothesis sets | -> abcd | inlier | ok]
  auto al = [](size_t v) { return (v + 63) & ~(size_t)63; };
  const size_t o_fs = 0, a_pts = al(sizeof(int) * (size_t)(P + 1)), a_sets = al(a_pts + sizeof(double) * 3 * (size_t)F);
  const size_t a_abcd = al(a_sets + sizeof(int) * sets.size()), a_inl = al(a_abcd + sizeof(double) * 4 * (size_t)P);
  const size_t a_ok = al(a_inl + (size_t)F), total = al(a_ok + (size_t)P);
  char *hb = nullptr, *blob = nullptr;
  {
    const int rca = ovp_io_arena(c, total, (void**)&hb, (void**)&blob);
    if (rca) return rca;
  }
  memcpy(hb + o_fs, b->feat_start, sizeof(int) * (size_t)(P + 1));
  memcpy(hb + a_pts, b->p_FinG, sizeof(double) * 3 * (size_t)F);
  memcpy(hb + a_sets, sets.data(), sizeof(int) * sets.size());
  PF_HIPCHK(hipMemcpyAsync(blob, hb, a_abcd, hipMemcpyHostToDevice, s));
  ovp::RansacJob j;
  j.feat_start = (const int*)(blob + o_fs);
  j.pts = (const double*)(blob + a_pts);
  j.sets = (const int*)(blob + a_sets);
  j.min_inlier_num = b->min_inlier_num;
  j.max_cond = b->max_cond;
  j.abcd = (double*)(blob + a_abcd);
  j.inlier = (unsigned char*)(blob + a_inl);
  j.ok = (unsigned char*)(blob + a_ok);
  hipLaunchKernelGGL(ovp::k_plane_ransac, dim3(P), dim3(256), 0, s, j);
  PF_HIPCHK(hipGetLastError());
  PF_HIPCHK(hipMemcpyAsync(hb + a_abcd, blob + a_abcd, total - a_abcd, hipMemcpyDeviceToHost, s));
  PF_HIPCHK(hipStreamSynchronize(s));
  memcpy(abcd, hb + a_abcd, sizeof(double) * 4 * (size_t)P);
  memcpy(inlier, hb + a_inl, (size_t)F);
  memcpy(ok, hb + a_ok, (size_t)P);
  return 0;
}

extern "C" int ovp_plane_optimize(ovp_ctx* c, const ovp_planeopt_batch* b, double* cp_out, double* p_out, uint8_t* kept,
                                  uint8_t* ok, int* iterations) {
  if (!c || !b || !cp_out || !p_out || !kept || !ok) return OVP_E_ARG;
  const int P = b->n_planes;
  if (P <= 0) return 0;
  const int F = b->feat_start[P], O = b->n_obs_total;
  int nf_max = 0;
  for (int pl = 0; pl < P; ++pl) nf_max = std::max(nf_max, b->feat_start[pl + 1] - b->feat_start[pl]);
  if (nf_max > 256) return OVP_E_CAPACITY;  // one thread per feature of a plane
  void* strm = nullptr;
  if (ovp_ctx_stream(c, &strm)) return OVP_E_ARG;
  hipStream_t s = (hipStream_t)strm;
  char* blob = nullptr;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    const size_t o = off;
    off = (off + bytes + 63) & ~(size_t)63;
    return o;
  };
  const size_t o_fs = take(sizeof(int) * (size_t)(P + 1)), o_p0 = take(sizeof(double) * 3 * (size_t)F);
  const size_t o_os = take(sizeof(int) * (size_t)F), o_no = take(sizeof(int) * (size_t)F);
  const size_t o_uv = take(sizeof(double) * 2 * (size_t)O), o_R = take(sizeof(double) * 9 * (size_t)O);
  const size_t o_pc = take(sizeof(double) * 3 * (size_t)O), o_cp = take(sizeof(double) * 3 * (size_t)P);
  // (the observation list of every plane: validated and counted first)
  size_t n_items = 0;
  for (int f = 0; f < F; ++f) {
    const int m = b->n_obs[f];
    if (m < 0 || (m > 0 && (b->obs_start[f] < 0 || b->obs_start[f] + m > O))) return OVP_E_ARG;
    n_items += (size_t)m;
  }
  const size_t o_fix = take((size_t)P);
  const size_t o_is = take(sizeof(int) * (size_t)(P + 1)), o_iob = take(sizeof(int) * (n_items + 1));
  const size_t o_ilf = take(sizeof(int) * (n_items + 1)), o_fi0 = take(sizeof(int) * (size_t)(F + 1));
  const size_t o_cpo = take(sizeof(double) * 3 * (size_t)P);
  const size_t o_po = take(sizeof(double) * 3 * (size_t)F), o_kept = take((size_t)F), o_ok = take((size_t)P);
  const size_t o_it = take(sizeof(int) * (size_t)P);
  const size_t in_bytes = o_cpo;  // everything in front of the outputs is input
  char* hb = nullptr;
  {
    const int rca = ovp_io_arena(c, off + 64, (void**)&hb, (void**)&blob);
    if (rca) return rca;
  }
#define PF_UP(o, src, bytes) memcpy(hb + (o), (src), (bytes))
  PF_UP(o_fs, b->feat_start, sizeof(int) * (size_t)(P + 1));
  PF_UP(o_p0, b->p_FinG, sizeof(double) * 3 * (size_t)F);
  PF_UP(o_os, b->obs_start, sizeof(int) * (size_t)F);
  PF_UP(o_no, b->n_obs, sizeof(int) * (size_t)F);
  if (O > 0) {
    PF_UP(o_uv, b->uv_norm, sizeof(double) * 2 * (size_t)O);
    PF_UP(o_R, b->R_GtoC, sizeof(double) * 9 * (size_t)O);
    PF_UP(o_pc, b->p_CinG, sizeof(double) * 3 * (size_t)O);
  }
  PF_UP(o_cp, b->cp, sizeof(double) * 3 * (size_t)P);
  PF_UP(o_fix, b->fix_plane, (size_t)P);
#undef PF_UP
  {
    int* is = (int*)(hb + o_is);
    int* iob = (int*)(hb + o_iob);
    int* ilf = (int*)(hb + o_ilf);
    int* fi0 = (int*)(hb + o_fi0);
    int it = 0;
    for (int pl = 0; pl < P; ++pl) {
      is[pl] = it;
      for (int f = b->feat_start[pl]; f < b->feat_start[pl + 1]; ++f) {
        fi0[f] = it - is[pl];
        for (int k = 0; k < b->n_obs[f]; ++k) {
          iob[it] = b->obs_start[f] + k;
          ilf[it] = f - b->feat_start[pl];
          ++it;
        }
      }
    }
    is[P] = it;
  }
  PF_HIPCHK(hipMemcpyAsync(blob, hb, in_bytes, hipMemcpyHostToDevice, s));
  ovp::RefineJob j;
  j.feat_start = (const int*)(blob + o_fs);
  j.p0 = (const double*)(blob + o_p0);
  j.obs_start = (const int*)(blob + o_os);
  j.n_obs = (const int*)(blob + o_no);
  j.uv = (const double*)(blob + o_uv);
  j.Rc = (const double*)(blob + o_R);
  j.pc = (const double*)(blob + o_pc);
  j.cp0 = (const double*)(blob + o_cp);
  j.fix_plane = (const unsigned char*)(blob + o_fix);
  j.item_start = (const int*)(blob + o_is);
  j.item_ob = (const int*)(blob + o_iob);
  j.item_lf = (const int*)(blob + o_ilf);
  j.feat_item0 = (const int*)(blob + o_fi0);
  j.sigma_px_norm = b->sigma_px_norm;
  j.sigma_c = b->sigma_c;
  // current camera: R_GtoC = R_ItoC R_GtoI, p_CinG = p_IinG - R_GtoC^T p_IinC   (PlaneFitting.cpp:444-453)
  for (int i = 0; i < 3; ++i)
    for (int k = 0; k < 3; ++k) {
      double sum = 0.0;
      for (int q = 0; q < 3; ++q) sum += b->R_ItoC[3 * i + q] * b->R_GtoI[3 * q + k];
      j.R_GtoC[3 * i + k] = sum;
    }
  for (int i = 0; i < 3; ++i) {
    double sum = 0.0;
    for (int q = 0; q < 3; ++q) sum += j.R_GtoC[3 * q + i] * b->p_IinC[q];
    j.p_CinG[i] = b->p_IinG[i] - sum;
  }
  j.cp_out = (double*)(blob + o_cpo);
  j.p_out = (double*)(blob + o_po);
  j.kept = (unsigned char*)(blob + o_kept);
  j.ok = (unsigned char*)(blob + o_ok);
  j.iterations = (int*)(blob + o_it);
  // (a full workgroup whatever the number of features: the threads beyond them take observations in the evaluations)
  hipLaunchKernelGGL(ovp::k_plane_refine, dim3(P), dim3(256), 0, s, j);
  PF_HIPCHK(hipGetLastError());
  PF_HIPCHK(hipMemcpyAsync(hb + o_cpo, blob + o_cpo, off - o_cpo, hipMemcpyDeviceToHost, s));
  PF_HIPCHK(hipStreamSynchronize(s));
#ifdef OVP_PF_STAMPS
  {
    long long h[64 * 12];
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(ovp::g_pf_stamps), sizeof(h));
    for (int pl = 0; pl < P && pl < 64; ++pl)
      fprintf(stderr,
              "refine plane %d: nf %lld items %lld iterations %lld | eval+J %lld  eval cost %lld  scale %lld  cauchy %lld  solve %lld (%lld "
              "tries)  dogleg %lld  total %lld (clock ticks)\n",
              pl, h[8 * pl + 4], h[8 * pl + 5], h[8 * pl + 3], h[8 * pl], h[8 * pl + 1], h[512 + 4 * pl], h[512 + 4 * pl + 1],
              h[512 + 4 * pl + 2], h[8 * pl + 6], h[512 + 4 * pl + 3], h[8 * pl + 2]);
  }
#endif
  memcpy(cp_out, hb + o_cpo, sizeof(double) * 3 * (size_t)P);
  memcpy(p_out, hb + o_po, sizeof(double) * 3 * (size_t)F);
  memcpy(kept, hb + o_kept, (size_t)F);
  memcpy(ok, hb + o_ok, (size_t)P);
  if (iterations) memcpy(iterations, hb + o_it, sizeof(int) * (size_t)P);
  return 0;
}
