// Plane path: UpdaterMSCKF::update's per-plane loop (update/UpdaterMSCKF.cpp:411-649) in information-pair form.
//
// For every plane (ascending id, each one a sequential EKF update at the state left by the previous plane):
//   per on-plane MSCKF feature: get_feature_jacobian_full with the point-on-plane rows (update/UpdaterHelper.cpp:448-512),
//   UpdaterPlane::nullspace_project_inplace (update/UpdaterPlane.cpp:483-517); then the stack is compressed with H_cp
//   carried along (:519-552), the plane is appended (in state) or projected out (not in state,
//   update/UpdaterMSCKF.cpp:593-604), a plane-level chi2 decides, and StateHelper::EKFUpdate is applied.
// Facts used (verified numerically):
//   * the m identical constraint rows of a feature (:503-511) are one row scaled by sqrt(m) as far as any Gram
//     product is concerned;
//   * H_cp lies in range(H_x) (gauge freedom), so the truncation after compression loses no information on
//     (x, cp): the retained system has exactly the Gram pair of the untruncated one;
//   * projecting out an out-of-state plane is the 3x3 Schur complement on that pair;
//   * there is no per-feature gate for plane features - only the plane-level one.
// So the kernels below only emit sparse rows, projector rows G = Q1^T [H_x | H_cp | r] and the constraint-row
// moments; K2's kernels reduce them; the update reuses K3 with a dense (non-triangular) factor M, P = M M^T,
// which is chained M <- M Lt^-T from plane to plane without re-factorizing P.
#include "ovp_feat_model.h"
#include "k_tile_body.h"
#include <utility>

namespace ovp {

typedef double double4_t __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// K1p: one wave per on-plane feature. lanes 0..2m-1 bearing rows; the merged constraint row is wave-uniform (an extra term of the sums).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_plane_feat(const FeatParams p, const PlaneParams pp) {
  const int fl = blockIdx.x;                 // local feature index within this plane
  const int f = pp.feat_list[fl];            // index into the feature batch
  const int lane = threadIdx.x;
  const int m = p.n_meas[f];
  const int n = 2 * m;
  const int a = lane >> 1, r = lane & 1;
  const bool valid = lane < n;
  const int* cidx = p.clone_idx + (size_t)f * p.max_meas;
  const int ci = cidx[valid ? a : 0];
  const int ida = p.clone_id[ci];
  __shared__ __attribute__((aligned(16))) double Gst[3 * OVP_LDG_CAP];

  double jrow[6], crow[14], hf[3], res;
  build_bearing_row(p, f, a, r, valid, ci, jrow, crow, hf, res);

  // constraint row (update/UpdaterHelper.cpp:450-497), merged: m identical rows == one row scaled by sqrt(m).  Its entries
  // depend on the feature and the plane only, not on an observation: every lane computes them (wave-uniform) and the row
  // takes part in the sums below as an extra term instead of occupying lane 2m - a track of 32 observations (64 bearing rows)
  // keeps its constraint (round 4; until then 2m + 1 rows had to fit the wavefront: OVP_E_CAPACITY above 31 observations,
  // update/UpdaterMSCKF.cpp:413-649 has no such limit)
  double hcp[3], hfc[3], resc;
  {
    const double* cp = pp.cp + 3 * pp.plane;
    const double* cpf = pp.in_state ? (pp.cp_fej + 3 * pp.plane) : cp;  // UpdaterMSCKF.cpp:467-475
    const double pf0 = p.p_FinG[3 * f], pf1 = p.p_FinG[3 * f + 1], pf2 = p.p_FinG[3 * f + 2];
    const double sm = sqrt((double)m) * pp.white_c;
    double d = sqrt(cp[0] * cp[0] + cp[1] * cp[1] + cp[2] * cp[2]);
    double id = 1.0 / d;
    double n0 = cp[0] * id, n1 = cp[1] * id, n2 = cp[2] * id;
    resc = sm * (0.0 - (n0 * pf0 + n1 * pf1 + n2 * pf2 - d));
    if (p.do_fej) {
      d = sqrt(cpf[0] * cpf[0] + cpf[1] * cpf[1] + cpf[2] * cpf[2]);
      id = 1.0 / d;
      n0 = cpf[0] * id;
      n1 = cpf[1] * id;
      n2 = cpf[2] * id;
    }
    const double ndp = n0 * pf0 + n1 * pf1 + n2 * pf2;  // p_FinG_fej == p_FinG for MSCKF features
    const double s = sm * id;
    hcp[0] = s * (pf0 - ndp * n0 - d * n0);
    hcp[1] = s * (pf1 - ndp * n1 - d * n1);
    hcp[2] = s * (pf2 - ndp * n2 - d * n2);
    hfc[0] = sm * n0;
    hfc[1] = sm * n1;
    hfc[2] = sm * n2;
  }

  // Q1 by CholeskyQR2 on H_f (2m+1 rows)
  double q[3] = {hf[0], hf[1], hf[2]};
  double qc[3] = {hfc[0], hfc[1], hfc[2]};  // the constraint row's part of Q1 (wave-uniform)
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    double v[8];
    v[0] = q[0] * q[0];
    v[1] = q[0] * q[1];
    v[2] = q[0] * q[2];
    v[3] = q[1] * q[1];
    v[4] = q[1] * q[2];
    v[5] = q[2] * q[2];
    v[6] = 0.0;
    v[7] = 0.0;
    const double rsum = wave_transpose_reduce<8>(v);
    const double g00 = readlane_f64(rsum, reduce_owner_lane<8>(0)) + qc[0] * qc[0], g01 = readlane_f64(rsum, reduce_owner_lane<8>(1)) + qc[0] * qc[1];
    const double g02 = readlane_f64(rsum, reduce_owner_lane<8>(2)) + qc[0] * qc[2], g11 = readlane_f64(rsum, reduce_owner_lane<8>(3)) + qc[1] * qc[1];
    const double g12 = readlane_f64(rsum, reduce_owner_lane<8>(4)) + qc[1] * qc[2], g22 = readlane_f64(rsum, reduce_owner_lane<8>(5)) + qc[2] * qc[2];
    // R^T R = G with the reciprocals of the diagonal (v_rsq_f64 + two Newton steps each): a sqrt and five divisions per pass were
    // 3.2 K cycles of dependent div / sqrt sequences.  Q1 is orthonormal to rounding after the second pass either way.
    const double i00 = rsqrt_nr2(g00), r01 = g01 * i00, r02 = g02 * i00;
    const double i11 = rsqrt_nr2(g11 - r01 * r01), r12 = (g12 - r01 * r02) * i11;
    const double i22 = rsqrt_nr2(g22 - r02 * r02 - r12 * r12);
    const double a0 = q[0] * i00;
    const double a1 = (q[1] - a0 * r01) * i11;
    const double a2 = (q[2] - a0 * r02 - a1 * r12) * i22;
    q[0] = a0;
    q[1] = a1;
    q[2] = a2;
    const double c0 = qc[0] * i00;
    const double c1 = (qc[1] - c0 * r01) * i11;
    const double c2 = (qc[2] - c0 * r02 - c1 * r12) * i22;
    qc[0] = c0;
    qc[1] = c1;
    qc[2] = c2;
  }

  const int ldg = p.ldg;
  for (int idx = lane; idx < 3 * ldg; idx += 64) Gst[idx] = 0.0;
  __syncthreads();
  double gsq = 0.0;  // |g|^2
  {
    double v[64];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int k = 0; k < 14; ++k) v[t * 14 + k] = q[t] * crow[k];
#pragma unroll
    for (int t = 0; t < 3; ++t) v[42 + t] = q[t] * res;
    v[45] = res * res;
#pragma unroll
    for (int k = 46; k < 64; ++k) v[k] = 0.0;
    double rsum = wave_transpose_reduce<64>(v);
    // the constraint row's terms of Q1^T r (lanes 42..44) and of r^T r (lane 45); its calibration / clone entries are zero
    rsum += lane == 42 ? qc[0] * resc : (lane == 43 ? qc[1] * resc : (lane == 44 ? qc[2] * resc : (lane == 45 ? resc * resc : 0.0)));
    {
      // where this lane's sum goes: lanes 0..41 = (projector row t, calibration entry k), 42..44 = the residual column.  The column
      // is picked by a select chain over the table - a kernel-argument array indexed by a lane-dependent k is a waterfall loop
      const int t = lane < 42 ? lane / 14 : lane - 42, k = lane - 14 * t;
      int col = -1;
#pragma unroll
      for (int kk = 0; kk < 14; ++kk) col = (k == kk && ((p.calmask >> kk) & 1)) ? p.calcol[kk] : col;
      if (lane >= 42) col = lane < 45 ? p.n : -1;
      if (col >= 0) Gst[t * ldg + col] = rsum;
    }
    const double g0 = readlane_f64(rsum, 42), g1 = readlane_f64(rsum, 43), g2 = readlane_f64(rsum, 44);
    gsq = g0 * g0 + g1 * g1 + g2 * g2;
    const double rr = readlane_f64(rsum, 45);
    // constraint-row moments (already scaled by m): hh (6), h*res (3), projected residual energy
    const double h0 = hcp[0], h1 = hcp[1], h2 = hcp[2];
    const double rc = resc;
    if (lane == 0) {
      double* o = pp.cst + (size_t)fl * 10;
      o[0] = h0 * h0;
      o[1] = h0 * h1;
      o[2] = h0 * h2;
      o[3] = h1 * h1;
      o[4] = h1 * h2;
      o[5] = h2 * h2;
      o[6] = h0 * rc;
      o[7] = h1 * rc;
      o[8] = h2 * rc;
      o[9] = rr - gsq;
    }
    // plane columns of G: only the constraint lane has a non-zero H_cp row
    const double qc0 = qc[0], qc1 = qc[1], qc2 = qc[2];
    if (lane < 9) {
      const int t = lane / 3, k = lane - 3 * t;
      const double qt = t == 0 ? qc0 : (t == 1 ? qc1 : qc2);
      const double hk = k == 0 ? h0 : (k == 1 ? h1 : h2);
      const int col = pp.in_state ? (pp.plane_sid + k) : (p.n + 1 + k);
      Gst[t * ldg + col] = qt * hk;
    }
  }
  // clone columns
  {
    double cv[18];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int l = 0; l < 6; ++l) {
        const double v = q[t] * jrow[l];
        cv[6 * t + l] = v + shfl_xor_f64(v, 1);
      }
    if (valid && r == 0) {  // (one masked block: eighteen of them were eighteen EXEC round trips)
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int l = 0; l < 6; ++l) Gst[t * ldg + ida + l] = cv[6 * t + l];
    }
  }
  __syncthreads();
  {
    double* gout = p.G + (size_t)3 * fl * ldg;
    for (int idx = lane; idx < 3 * ldg; idx += 64) gout[idx] = Gst[idx];
  }
  // sparse bearing rows, grouped by clone slot, local feature index
  // clone slots this feature was seen from: every valid lane already holds its observation's slot (a rolled loop over cidx[]
  // was m dependent loads, ~6 us of a 12 us kernel)
  const unsigned long long seen = wave_or_u64(valid ? (1ull << ci) : 0ull);
  if (valid) {
    double* ro = p.rec + (((size_t)ci * pp.n_local + fl) * 2 + r) * OVP_REC;
#pragma unroll
    for (int l = 0; l < 6; ++l) ro[l] = jrow[l];
#pragma unroll
    for (int k = 0; k < 14; ++k) ro[6 + k] = crow[k];
    ro[20] = res;
  }
  for (int cc = 0; cc < p.n_clones; ++cc) {
    if (!((seen >> cc) & 1ull)) {
      double* ro = p.rec + (((size_t)cc * pp.n_local + fl) * 2) * OVP_REC;
      if (lane < 2 * OVP_REC) ro[lane] = 0.0;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// cstsum[10] = sum over the plane's features of cst (fixed order tree)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_reduce_cst(const double* __restrict__ cst, int nf, double* __restrict__ out) {
  __shared__ double red[256];
  const int t = threadIdx.x, e = t % 10, lane = t / 10;  // 25 partial sums per element
  double s = 0.0;
  if (t < 250)
    for (int f = lane; f < nf; f += 25) s += cst[(size_t)f * 10 + e];
  red[t] = (t < 250) ? s : 0.0;
  __syncthreads();
  if (t < 10) {
    double acc = 0.0;
    for (int l = 0; l < 25; ++l) acc += red[l * 10 + t];
    out[t] = acc;
  }
}

// ------------------------------------------------------------------------------------------------
// Extended assembly over the column space of G: 0..n-1 state, n residual, n+1..n+3 out-of-state plane.
// E[(n+4)][lde]; plane columns take their structured part from cstsum.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int gram_index2(int p, int q) { return p * OVP_REC - (p * (p - 1)) / 2 + (q - p); }

__global__ __launch_bounds__(256) void k_assemble_ext(const double* __restrict__ gramR, int n_clones,
                                                       const double* __restrict__ part, int n_split, int ntile,
                                                       const ColMap* __restrict__ colmap, int n, int plane_sid,
                                                       const double* __restrict__ cstsum, double* __restrict__ E,
                                                       int lde) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  const int r = blockIdx.y;
  const int ne = n + 4;
  if (c >= ne) return;
  auto classify = [&](int col) {
    ColMap m;
    m.kind = 0;
    m.idx = m.off = m.pad = 0;
    if (col < n) {
      m = colmap[col];
      if (plane_sid >= 0 && col >= plane_sid && col < plane_sid + 3) {
        m.kind = 4;
        m.idx = col - plane_sid;
      }
    } else if (col == n) {
      m.kind = 3;
    } else {
      if (plane_sid < 0) {
        m.kind = 4;
        m.idx = col - n - 1;
      }
    }
    return m;
  };
  const ColMap mr = classify(r), mc = classify(c);
  double s = 0.0;
  if (mr.kind == 4 || mc.kind == 4) {
    if (mr.kind == 4 && mc.kind == 4) {
      const int i = min(mr.idx, mc.idx), j = max(mr.idx, mc.idx);
      const int e = i == 0 ? j : (i == 1 ? 2 + j : 5);  // (0,0)(0,1)(0,2)(1,1)(1,2)(2,2)
      s = cstsum[e];
    } else if (mr.kind == 3 || mc.kind == 3) {
      s = cstsum[6 + (mr.kind == 4 ? mr.idx : mc.idx)];
    }
  } else if (mr.kind != 0 && mc.kind != 0) {
    auto gcol = [](const ColMap& m) { return m.kind == 1 ? m.off : (m.kind == 2 ? 6 + m.idx : 20); };
    const int gr = gcol(mr), gc = gcol(mc);
    const int p = min(gr, gc), q = max(gr, gc);
    const int gi = gram_index2(p, q);
    if (mr.kind == 1 && mc.kind == 1) {
      if (mr.idx == mc.idx) s = gramR[(size_t)mr.idx * OVP_GRAM_ELEMS + gi];
    } else if (mr.kind == 1) {
      s = gramR[(size_t)mr.idx * OVP_GRAM_ELEMS + gi];
    } else if (mc.kind == 1) {
      s = gramR[(size_t)mc.idx * OVP_GRAM_ELEMS + gi];
    } else {
#pragma unroll 8
      for (int sl = 0; sl < n_clones; ++sl) s += gramR[(size_t)sl * OVP_GRAM_ELEMS + gi];
    }
  }
  {
    const int I = max(r, c), J = min(r, c);
    const int ti = I >> 4, tj = J >> 4;
    const int tile = ti * (ti + 1) / 2 + tj;
    const int e = (I & 15) * 16 + (J & 15);
    double d = 0.0;
#pragma unroll 4
    for (int sp = 0; sp < n_split; ++sp) d += part[((size_t)sp * ntile + tile) * 256 + e];
    s -= d;
  }
  E[(size_t)r * lde + c] = s;
}

// ------------------------------------------------------------------------------------------------
// SLAM landmarks on a plane that is not in the state (update/UpdaterMSCKF.cpp:232-252, :545-552): one point-on-plane row per
// landmark (update/UpdaterHelper.cpp:448-505), added to the extended pair E over [state | residual | plane]:
//   h = [ w n^T at the landmark's columns | w/d (p - (n.p) n - d n)^T at the plane's ],  r = -w (n.p - d)
// Jacobians at the first estimates when do_fej.  One workgroup, the rows one after the other (there are a handful).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_plane_slam_rows(double* __restrict__ E, int lde, int n, int plane1, int n_slam,
                                                         const int* __restrict__ slam_plane, const int* __restrict__ slam_id,
                                                         const double* __restrict__ slam_p, const double* __restrict__ slam_p_fej,
                                                         const double* __restrict__ cp, const double* __restrict__ cp_fej,
                                                         double white_c, int do_fej, double* __restrict__ cstsum) {
  __shared__ double h[7];
  __shared__ int col[7];
  const int t = threadIdx.x;
  for (int q = 0; q < n_slam; ++q) {
    if (slam_plane[q] != plane1) continue;  // uniform
    if (t == 0) {
      const double* pv = slam_p + 3 * q;
      const double* pj = do_fej ? slam_p_fej + 3 * q : pv;
      const double* cj = do_fej ? cp_fej : cp;
      double d = sqrt(cp[0] * cp[0] + cp[1] * cp[1] + cp[2] * cp[2]);
      double nv[3] = {cp[0] / d, cp[1] / d, cp[2] / d};
      const double r = white_c * (0.0 - ((nv[0] * pv[0] + nv[1] * pv[1] + nv[2] * pv[2]) - d));
      d = sqrt(cj[0] * cj[0] + cj[1] * cj[1] + cj[2] * cj[2]);
      nv[0] = cj[0] / d;
      nv[1] = cj[1] / d;
      nv[2] = cj[2] / d;
      const double np = nv[0] * pj[0] + nv[1] * pj[1] + nv[2] * pj[2];
      for (int k = 0; k < 3; ++k) {
        h[k] = white_c * nv[k];
        h[3 + k] = white_c * 1.0 / d * (pj[k] - np * nv[k] - d * nv[k]);
        col[k] = slam_id[q] + k;
        col[3 + k] = n + 1 + k;
      }
      h[6] = r;
      col[6] = n;
      cstsum[9] += r * r;
    }
    __syncthreads();
    if (t < 49) {
      const int i = t / 7, k = t - 7 * i;
      if (!(i == 6 && k == 6)) E[(size_t)col[i] * lde + col[k]] += h[i] * h[k];
    }
    __syncthreads();
  }
}

// landmarks are state variables: ext Vec::update after an accepted plane
__global__ void k_plane_commit_slam(const double* __restrict__ res, const double* __restrict__ dx, int n_slam,
                                    const int* __restrict__ slam_id, double* __restrict__ slam_p) {
  if (!(res[1] > 0.5)) return;
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n_slam) return;
  for (int k = 0; k < 3; ++k) slam_p[3 * q + k] += dx[slam_id[q] + k];
}

// ------------------------------------------------------------------------------------------------
// E -> Ab (n+1 rows): in-state plane: copy; out-of-state: Schur complement on columns n+1..n+3
// (== UpdaterHelper::nullspace_project_inplace(Hcp_big, Hx_big, res_big), update/UpdaterMSCKF.cpp:603).
// scal[0] = projected residual energy rr (input: cstsum[9] + bearing energy is folded by the caller).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_plane_reduce_to_state(const double* __restrict__ E, int lde, int n,
                                                                int in_state, double* __restrict__ Ab, int lda,
                                                                const double* __restrict__ rr_in,
                                                                double* __restrict__ scal) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  const int r = blockIdx.y;  // 0..n (n = b row)
  // 3x3 inverse of A_cc (every thread, it is tiny)
  double Ai[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  double bc[3] = {0, 0, 0};
  if (!in_state) {
    const double a00 = E[(size_t)(n + 1) * lde + n + 1], a01 = E[(size_t)(n + 1) * lde + n + 2];
    const double a02 = E[(size_t)(n + 1) * lde + n + 3], a11 = E[(size_t)(n + 2) * lde + n + 2];
    const double a12 = E[(size_t)(n + 2) * lde + n + 3], a22 = E[(size_t)(n + 3) * lde + n + 3];
    const double c00 = a11 * a22 - a12 * a12, c01 = a02 * a12 - a01 * a22, c02 = a01 * a12 - a02 * a11;
    const double det = a00 * c00 + a01 * c01 + a02 * c02;
    const double id = 1.0 / det;
    Ai[0] = c00 * id;
    Ai[1] = c01 * id;
    Ai[2] = c02 * id;
    Ai[3] = Ai[1];
    Ai[4] = (a00 * a22 - a02 * a02) * id;
    Ai[5] = (a01 * a02 - a00 * a12) * id;
    Ai[6] = Ai[2];
    Ai[7] = Ai[5];
    Ai[8] = (a00 * a11 - a01 * a01) * id;
    bc[0] = E[(size_t)n * lde + n + 1];
    bc[1] = E[(size_t)n * lde + n + 2];
    bc[2] = E[(size_t)n * lde + n + 3];
  }
  if (c == 0 && r == 0) {
    double rr = rr_in[0];
    if (!in_state)
      rr -= bc[0] * (Ai[0] * bc[0] + Ai[1] * bc[1] + Ai[2] * bc[2]) + bc[1] * (Ai[3] * bc[0] + Ai[4] * bc[1] + Ai[5] * bc[2]) +
            bc[2] * (Ai[6] * bc[0] + Ai[7] * bc[1] + Ai[8] * bc[2]);
    scal[0] = rr;
  }
  if (c >= n) return;
  double v = E[(size_t)r * lde + c];
  if (!in_state) {
    const double xr0 = E[(size_t)r * lde + n + 1], xr1 = E[(size_t)r * lde + n + 2], xr2 = E[(size_t)r * lde + n + 3];
    const double xc0 = E[(size_t)(n + 1) * lde + c], xc1 = E[(size_t)(n + 2) * lde + c], xc2 = E[(size_t)(n + 3) * lde + c];
    const double t0 = Ai[0] * xc0 + Ai[1] * xc1 + Ai[2] * xc2;
    const double t1 = Ai[3] * xc0 + Ai[4] * xc1 + Ai[5] * xc2;
    const double t2 = Ai[6] * xc0 + Ai[7] * xc1 + Ai[8] * xc2;
    v -= xr0 * t0 + xr1 * t1 + xr2 * t2;
  }
  Ab[(size_t)r * lda + c] = v;
}

// total projected residual energy of the plane = bearing rows (element (20,20) of the per-clone Grams) +
// constraint rows (cstsum[9] holds sum_f (|res_f|^2 - |g_f|^2) already, computed in-wave) -> rr[0] = cstsum[9]
__global__ void k_copy_scalar(const double* __restrict__ src, double* __restrict__ dst) {
  if (threadIdx.x == 0 && blockIdx.x == 0) dst[0] = src[0];
}

// ------------------------------------------------------------------------------------------------
// Range part of the residual: pr = b^T A^+ b via a diagonally normalised, regularised system
//   An = D^-1/2 A D^-1/2 + eps I,  bn = D^-1/2 b   (columns with A_ii <= 0 are not involved: unit pivot, bn = 0)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_normalize_reg(const double* __restrict__ Ab, int lda, int n, double eps,
                                                        double* __restrict__ An, double* __restrict__ bn) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  const int r = blockIdx.y;
  if (c >= n) return;
  const double dr = Ab[(size_t)r * lda + r], dc = Ab[(size_t)c * lda + c];
  double v;
  if (dr > 0.0 && dc > 0.0) {
    v = Ab[(size_t)r * lda + c] / sqrt(dr * dc);
    if (r == c) v = 1.0 + eps;
  } else {
    v = (r == c) ? 1.0 : 0.0;
  }
  An[(size_t)r * lda + c] = v;
  if (r == 0) {
    bn[c] = dc > 0.0 ? Ab[(size_t)n * lda + c] / sqrt(dc) : 0.0;
  }
}

// y = Lr^-1 bn (single workgroup, blocked with the inverse diagonal blocks), pr = |y|^2, n_deg = #pivots^2 < tol
__global__ __launch_bounds__(256) void k_range_energy(const double* __restrict__ Lr, const double* __restrict__ Dinv,
                                                       const double* __restrict__ bn, int n, int ld, double tol,
                                                       double* __restrict__ scal /* [1]=pr, [2]=n_deg */) {
  extern __shared__ double y[];  // n rounded up to 16
  __shared__ double tmp[16];
  const int t = threadIdx.x;
  const int nt = (n + 15) >> 4;
  for (int i = t; i < nt * 16; i += 256) y[i] = 0.0;
  __syncthreads();
  for (int i = 0; i < nt; ++i) {
    // tmp = b_i - sum_{k<i} L_ik y_k   (16 rows x 16 partial lanes)
    const int row = t >> 4, part = t & 15;
    const int gr = 16 * i + row;
    double s = 0.0;
    if (gr < n)
      for (int cidx = part; cidx < 16 * i; cidx += 16) s = fma(Lr[(size_t)gr * ld + cidx], y[cidx], s);
    s += shfl_xor_f64(s, 8);
    s += shfl_xor_f64(s, 4);
    s += shfl_xor_f64(s, 2);
    s += shfl_xor_f64(s, 1);
    if (part == 0) tmp[row] = (gr < n ? bn[gr] : 0.0) - s;
    __syncthreads();
    if (t < 16) {
      const double* di = Dinv + (size_t)i * 256 + t * 16;
      double acc = 0.0;
#pragma unroll
      for (int k = 0; k < 16; ++k) acc = fma(di[k], tmp[k], acc);
      y[16 * i + t] = acc;
    }
    __syncthreads();
  }
  // reductions
  double pr = 0.0, nd = 0.0;
  for (int i = t; i < n; i += 256) {
    pr += y[i] * y[i];
    const double piv = Lr[(size_t)i * ld + i];
    if (piv * piv < tol) nd += 1.0;
  }
  __shared__ double r1[256], r2[256];
  r1[t] = pr;
  r2[t] = nd;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (t < s) {
      r1[t] += r1[t + s];
      r2[t] += r2[t + s];
    }
    __syncthreads();
  }
  if (t == 0) {
    scal[1] = r1[0];
    scal[2] = r2[0];
  }
}

// u = V b ; dx = V^T u ; bdx = b . dx          (one workgroup; n <= 1024)
__global__ __launch_bounds__(1024) void k_dx_from_factor(const double* __restrict__ V, int n, int ld,
                                                          const double* __restrict__ b, double* __restrict__ dx,
                                                          double* __restrict__ scal /* [3] = b.dx */) {
  extern __shared__ double sh[];  // u[n], red[1024]
  double* u = sh;
  double* red = sh + n;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  for (int row = wave; row < n; row += 16) {
    const double* vr = V + (size_t)row * ld;
    double s = 0.0;
    for (int c = lane; c < n; c += 64) s = fma(vr[c], b[c], s);
    s = wave_sum(s);
    if (lane == 0) u[row] = s;
  }
  __syncthreads();
  double bd = 0.0;
  for (int c = t; c < n; c += 1024) {
    double s = 0.0;
    for (int row = 0; row < n; ++row) s = fma(V[(size_t)row * ld + c], u[row], s);
    dx[c] = s;
    bd = fma(b[c], s, bd);
  }
  red[t] = bd;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if (t < s) red[t] += red[t + s];
    __syncthreads();
  }
  if (t == 0) scal[3] = red[0];
}

// ------------------------------------------------------------------------------------------------
// plane-level gate (update/UpdaterMSCKF.cpp:606-631) and conditional commit (:646-648 + ext Type::update).
//   chi2 = (pr - b.dx) + (rows_u - rank) * (rr - pr) / (rows_live - rank)        [DESIGN.md §3b: deterministic stand-in for the
//   reference's rounding-dependent statistic], accept iff chi2 <= thr and the factorizations succeeded.
// On accept: M <- V^T, pose tables / calibration / in-state planes updated with dx, dx stored for the host.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_plane_gate(const double* __restrict__ scal, const int* __restrict__ flags,
                                                     double thr, int rows_live, int rows_u, int n_involved, int force,
                                                     double* __restrict__ res_out /* [4]: chi2, accept, n_deg, pr */) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const double rr = scal[0], pr = scal[1], ndeg = scal[2], bdx = scal[3];
  // rank of the retained system; pivots of non-involved columns are exactly 1 and never counted as deficient.
  // The reference keeps rows_u rows of which (rows_u - rank) carry no Jacobian, only residual noise.
  // They sample the rows_live = sum(2m - 2) directions that carry residual energy (k_chol2.hip, mode 1, for the argument).
  const double rank = (double)n_involved - ndeg;
  const double noise_rows = fmax((double)rows_u - rank, 0.0);
  const double denom = (double)rows_live - rank;
  const double frac = denom > 0.5 ? fmin(noise_rows / denom, 1.0) : 1.0;
  const double chi2 = (pr - bdx) + OVP_PLANE_NOISE_KAPPA * frac * fmax(rr - pr, 0.0);
  // force: 0 / 1 = decision handed over by the caller (ovp_plane_batch::force_decision), anything else = the gate decides
  const bool ok = (flags[0] == 0) && (force == 0 ? false : (force == 1 ? true : (chi2 <= thr)));
  res_out[0] = chi2;
  res_out[1] = ok ? 1.0 : 0.0;
  res_out[2] = ndeg;
  res_out[3] = pr;
}

__global__ __launch_bounds__(256) void k_plane_commit(const double* __restrict__ res, const double* __restrict__ V,
                                                       double* __restrict__ M, int n, int ld,
                                                       const double* __restrict__ dx, double* __restrict__ dx_out,
                                                       double* __restrict__ clone_R, double* __restrict__ clone_p,
                                                       const int* __restrict__ clone_id, int n_clones,
                                                       double* __restrict__ cal, int calib_id, int intr_id,
                                                       double* __restrict__ cp, const int* __restrict__ plane_sid,
                                                       int n_planes) {
  const bool ok = res[1] > 0.5;
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int nthreads = gridDim.x * blockDim.x;
  for (int i = gid; i < n; i += nthreads) dx_out[i] = ok ? dx[i] : 0.0;
  if (!ok) return;
  // M <- V^T (first-generation loop: the chained factor; V == nullptr when nothing is chained)
  for (int idx = gid; V && idx < n * n; idx += nthreads) {
    const int r = idx / n, c = idx - r * n;
    M[(size_t)r * ld + c] = V[(size_t)c * ld + r];
  }
  // ext JPLQuat::update on rotation matrices: R <- R(dq) R, dq = quatnorm([dth/2, 1]); positions additive
  auto rot_update = [&](double* R, const double* dth) {
    double qx = 0.5 * dth[0], qy = 0.5 * dth[1], qz = 0.5 * dth[2], qw = 1.0;
    const double nn = 1.0 / sqrt(qx * qx + qy * qy + qz * qz + qw * qw);
    qx *= nn;
    qy *= nn;
    qz *= nn;
    qw *= nn;
    const double a = 2.0 * qw * qw - 1.0;
    double D[9];
    D[0] = a + 2.0 * qx * qx;
    D[1] = 2.0 * qw * qz + 2.0 * qx * qy;
    D[2] = -2.0 * qw * qy + 2.0 * qx * qz;
    D[3] = -2.0 * qw * qz + 2.0 * qy * qx;
    D[4] = a + 2.0 * qy * qy;
    D[5] = 2.0 * qw * qx + 2.0 * qy * qz;
    D[6] = 2.0 * qw * qy + 2.0 * qz * qx;
    D[7] = -2.0 * qw * qx + 2.0 * qz * qy;
    D[8] = a + 2.0 * qz * qz;
    double O[9];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) O[3 * i + j] = D[3 * i] * R[j] + D[3 * i + 1] * R[3 + j] + D[3 * i + 2] * R[6 + j];
    for (int i = 0; i < 9; ++i) R[i] = O[i];
  };
  if (gid < n_clones) {
    const int id = clone_id[gid];
    rot_update(clone_R + 9 * gid, dx + id);
    for (int k = 0; k < 3; ++k) clone_p[3 * gid + k] += dx[id + 3 + k];
  } else if (gid == n_clones) {
    if (calib_id >= 0) {
      rot_update(cal, dx + calib_id);
      for (int k = 0; k < 3; ++k) cal[9 + k] += dx[calib_id + 3 + k];
    }
    if (intr_id >= 0)
      for (int k = 0; k < 8; ++k) cal[12 + k] += dx[intr_id + k];
  } else if (gid > n_clones && gid <= n_clones + n_planes) {
    const int pl = gid - n_clones - 1;
    if (plane_sid[pl] >= 0)
      for (int k = 0; k < 3; ++k) cp[3 * pl + k] += dx[plane_sid[pl] + k];
  }
}

// ------------------------------------------------------------------------------------------------
// Plane initialisation (UpdaterPlane::init_vio_plane -> StateHelper::initialize, update/UpdaterPlane.cpp:436-446,
// state/StateHelper.cpp:398-586) in information form: with a flat prior on the new plane the joint posterior of
// (x, cp) given the extended pair E = [[Exx Exc][Ecx Ecc]], [bx; bc] is
//   Sigma_xx = P+ (already resident),  Sigma_xc = -P+ Z,  Sigma_cc = Ecc^-1 + Z^T P+ Z,  Z = Exc Ecc^-1,
//   d cp = Ecc^-1 (bc - Ecx dx).
// One workgroup; P already holds P+ (n x n); rows/columns n..n+2 are appended.  The pair may live on a SELECTION of ns state columns
// (ids != nullptr: E is (ns + 4) wide, its row / column i stands for state column ids[i], every other column of the state has a zero
// row in Exc): the sums then run over the selection only.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_plane_init_augment(const double* __restrict__ E, int lde, int ns,
                                                             const int* __restrict__ ids, int n, double* __restrict__ P, int ldp,
                                                             const double* __restrict__ dx,
                                                             double* __restrict__ out /* [3] d cp */) {
  extern __shared__ double Z[];  // ns x 3
  __shared__ double Ai[9], red[256 * 3];
  const int t = threadIdx.x;
  if (t == 0) {
    const double a00 = E[(size_t)(ns + 1) * lde + ns + 1], a01 = E[(size_t)(ns + 1) * lde + ns + 2];
    const double a02 = E[(size_t)(ns + 1) * lde + ns + 3], a11 = E[(size_t)(ns + 2) * lde + ns + 2];
    const double a12 = E[(size_t)(ns + 2) * lde + ns + 3], a22 = E[(size_t)(ns + 3) * lde + ns + 3];
    const double c00 = a11 * a22 - a12 * a12, c01 = a02 * a12 - a01 * a22, c02 = a01 * a12 - a02 * a11;
    const double id = 1.0 / (a00 * c00 + a01 * c01 + a02 * c02);
    Ai[0] = c00 * id;
    Ai[1] = c01 * id;
    Ai[2] = c02 * id;
    Ai[3] = Ai[1];
    Ai[4] = (a00 * a22 - a02 * a02) * id;
    Ai[5] = (a01 * a02 - a00 * a12) * id;
    Ai[6] = Ai[2];
    Ai[7] = Ai[5];
    Ai[8] = (a00 * a11 - a01 * a01) * id;
  }
  __syncthreads();
  for (int r = t; r < ns; r += 256) {
    const double x0 = E[(size_t)r * lde + ns + 1], x1 = E[(size_t)r * lde + ns + 2], x2 = E[(size_t)r * lde + ns + 3];
    Z[3 * r + 0] = x0 * Ai[0] + x1 * Ai[3] + x2 * Ai[6];
    Z[3 * r + 1] = x0 * Ai[1] + x1 * Ai[4] + x2 * Ai[7];
    Z[3 * r + 2] = x0 * Ai[2] + x1 * Ai[5] + x2 * Ai[8];
  }
  __syncthreads();
  // Sigma_xc rows
  for (int r = t; r < n; r += 256) {
    const double* pr = P + (size_t)r * ldp;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int c = 0; c < ns; ++c) {
      const double pv = pr[ids ? ids[c] : c];
      s0 = fma(pv, Z[3 * c + 0], s0);
      s1 = fma(pv, Z[3 * c + 1], s1);
      s2 = fma(pv, Z[3 * c + 2], s2);
    }
    const double v[3] = {-s0, -s1, -s2};
    for (int k = 0; k < 3; ++k) {
      P[(size_t)r * ldp + n + k] = v[k];
      P[(size_t)(n + k) * ldp + r] = v[k];
    }
  }
  __syncthreads();  // (the rows just written are read back below, by other threads of this workgroup)
  // the partial sums of Z^T Sigma_xc (3x3) and Ecx dx (3): over the selection
  double acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = t; i < ns; i += 256) {
    const int r = ids ? ids[i] : i;
    double v[3];
    for (int k = 0; k < 3; ++k) v[k] = P[(size_t)(n + k) * ldp + r];
    for (int a = 0; a < 3; ++a)
      for (int j = 0; j < 3; ++j) acc[3 * a + j] += Z[3 * i + a] * v[j];
    const double dxr = dx[r];
    acc[9] += E[(size_t)i * lde + ns + 1] * dxr;
    acc[10] += E[(size_t)i * lde + ns + 2] * dxr;
    acc[11] += E[(size_t)i * lde + ns + 3] * dxr;
  }
  __shared__ double tot[12];
  for (int q = 0; q < 12; q += 3) {
    red[t] = acc[q];
    red[256 + t] = acc[q + 1];
    red[512 + t] = acc[q + 2];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (t < s) {
        red[t] += red[t + s];
        red[256 + t] += red[256 + t + s];
        red[512 + t] += red[512 + t + s];
      }
      __syncthreads();
    }
    if (t == 0) {
      tot[q] = red[0];
      tot[q + 1] = red[256];
      tot[q + 2] = red[512];
    }
    __syncthreads();
  }
  if (t < 9) {
    const int i = t / 3, j = t - 3 * i;
    // Sigma_cc = Ecc^-1 - Z^T Sigma_xc  (symmetrised)
    const double v = Ai[t] - 0.5 * (tot[3 * i + j] + tot[3 * j + i]);
    P[(size_t)(n + i) * ldp + n + j] = v;
  }
  if (t < 3) {
    const double b0 = E[(size_t)ns * lde + ns + 1] - tot[9], b1 = E[(size_t)ns * lde + ns + 2] - tot[10];
    const double b2 = E[(size_t)ns * lde + ns + 3] - tot[11];
    out[t] = Ai[3 * t] * b0 + Ai[3 * t + 1] * b1 + Ai[3 * t + 2] * b2;
  }
}

}  // namespace ovp

extern "C" {

hipError_t ovp_launch_plane_init_augment(const double* E, int lde, int ns, const int* ids, int n, double* P, int ldp, const double* dx,
                                         double* out, hipStream_t stream) {
  hipLaunchKernelGGL(ovp::k_plane_init_augment, dim3(1), dim3(256), (size_t)3 * ns * sizeof(double), stream, E, lde, ns, ids, n, P,
                     ldp, dx, out);
  return hipGetLastError();
}

hipError_t ovp_launch_plane_feat(const ovp::FeatParams* p, const ovp::PlaneParams* pp, int n_local,
                                 hipStream_t stream) {
  if (n_local <= 0) return hipSuccess;
  hipLaunchKernelGGL(ovp::k_plane_feat, dim3(n_local), dim3(64), 0, stream, *p, *pp);
  return hipGetLastError();
}
hipError_t ovp_launch_reduce_cst(const double* cst, int nf, double* out, hipStream_t stream) {
  hipLaunchKernelGGL(ovp::k_reduce_cst, dim3(1), dim3(256), 0, stream, cst, nf, out);
  return hipGetLastError();
}
hipError_t ovp_launch_assemble_ext(const double* gramR, int n_clones, const double* part, int n_split,
                                   const ovp::ColMap* colmap, int n, int plane_sid, const double* cstsum, double* E,
                                   int lde, hipStream_t stream) {
  const int nt = (n + 4 + 15) / 16;
  const int ntile = nt * (nt + 1) / 2;
  hipLaunchKernelGGL(ovp::k_assemble_ext, dim3((n + 4 + 255) / 256, n + 4), dim3(256), 0, stream, gramR, n_clones, part,
                     n_split, ntile, colmap, n, plane_sid, cstsum, E, lde);
  return hipGetLastError();
}
hipError_t ovp_launch_plane_reduce_to_state(const double* E, int lde, int n, int in_state, double* Ab, int lda,
                                            const double* rr_in, double* scal, hipStream_t stream) {
  hipLaunchKernelGGL(ovp::k_plane_reduce_to_state, dim3((n + 255) / 256, n + 1), dim3(256), 0, stream, E, lde, n,
                     in_state, Ab, lda, rr_in, scal);
  return hipGetLastError();
}
hipError_t ovp_launch_normalize_reg(const double* Ab, int lda, int n, double eps, double* An, double* bn,
                                    hipStream_t stream) {
  hipLaunchKernelGGL(ovp::k_normalize_reg, dim3((n + 255) / 256, n), dim3(256), 0, stream, Ab, lda, n, eps, An, bn);
  return hipGetLastError();
}
hipError_t ovp_launch_range_energy(const double* Lr, const double* Dinv, const double* bn, int n, int ld, double tol,
                                   double* scal, hipStream_t stream) {
  const size_t shmem = (size_t)(((n + 15) / 16) * 16) * sizeof(double);
  hipLaunchKernelGGL(ovp::k_range_energy, dim3(1), dim3(256), shmem, stream, Lr, Dinv, bn, n, ld, tol, scal);
  return hipGetLastError();
}
hipError_t ovp_launch_dx_from_factor(const double* V, int n, int ld, const double* b, double* dx, double* scal,
                                     hipStream_t stream) {
  const size_t shmem = (size_t)(n + 1024) * sizeof(double);
  hipLaunchKernelGGL(ovp::k_dx_from_factor, dim3(1), dim3(1024), shmem, stream, V, n, ld, b, dx, scal);
  return hipGetLastError();
}
hipError_t ovp_launch_plane_gate(const double* scal, const int* flags, double thr, int rows_live, int rows_u,
                                 int n_involved, int force, double* res_out, hipStream_t stream) {
  hipLaunchKernelGGL(ovp::k_plane_gate, dim3(1), dim3(64), 0, stream, scal, flags, thr, rows_live, rows_u, n_involved,
                     force, res_out);
  return hipGetLastError();
}
hipError_t ovp_launch_plane_slam_rows(double* E, int lde, int n, int plane1, int n_slam, const int* slam_plane, const int* slam_id,
                                      const double* slam_p, const double* slam_p_fej, const double* cp, const double* cp_fej,
                                      double white_c, int do_fej, double* cstsum, hipStream_t stream) {
  hipLaunchKernelGGL(ovp::k_plane_slam_rows, dim3(1), dim3(64), 0, stream, E, lde, n, plane1, n_slam, slam_plane, slam_id, slam_p,
                     slam_p_fej, cp, cp_fej, white_c, do_fej, cstsum);
  return hipGetLastError();
}

hipError_t ovp_launch_plane_commit_slam(const double* res, const double* dx, int n_slam, const int* slam_id, double* slam_p,
                                        hipStream_t stream) {
  hipLaunchKernelGGL(ovp::k_plane_commit_slam, dim3((n_slam + 63) / 64), dim3(64), 0, stream, res, dx, n_slam, slam_id, slam_p);
  return hipGetLastError();
}

hipError_t ovp_launch_plane_commit(const double* res, const double* V, double* M, int n, int ld, const double* dx,
                                   double* dx_out, double* clone_R, double* clone_p, const int* clone_id, int n_clones,
                                   double* cal, int calib_id, int intr_id, double* cp, const int* plane_sid,
                                   int n_planes, hipStream_t stream) {
  hipLaunchKernelGGL(ovp::k_plane_commit, dim3(64), dim3(256), 0, stream, res, V, M, n, ld, dx, dx_out, clone_R, clone_p,
                     clone_id, n_clones, cal, calib_id, intr_id, cp, plane_sid, n_planes);
  return hipGetLastError();
}
}
