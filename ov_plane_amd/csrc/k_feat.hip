// K1: per-feature build -> (implicit) left-nullspace projection -> chi2 gate -> projector rows.
//
// One wavefront per MSCKF point feature (lane i <-> measurement row i, lanes 2a and 2a+1 <-> observation a).
// Replaces the per-feature body of UpdaterMSCKF::update (update/UpdaterMSCKF.cpp:695-786):
//   get_feature_jacobian_full   update/UpdaterHelper.cpp:195-513  (GLOBAL_3D, radtan, mono)
//   nullspace_project_inplace   update/UpdaterHelper.cpp:515-546
//   gate                        update/UpdaterMSCKF.cpp:739-764
// MI355X-first formulation: the stacked Jacobian is never densified.  Row pair a of H_x only
// touches clone(a)'s 6 columns and the 14 calibration columns, so
//   B = H_x P H_x^T + I                      is built from 6x6 / 6x14 / 14x14 blocks of P,
//   chi2 = r^T N (N^T B N)^-1 N^T r          = y^T y - (Z^T y)^T (Z^T Z)^-1 (Z^T y),  L L^T = B, y = L^-1 r, Z = L^-1 H_f
// (N = left nullspace of H_f; identity valid for any orthonormal N), and the feature's contribution to the
// information pair is  Hp^T Hp = H_x^T H_x - G^T G,  Hp^T rp = H_x^T r - G^T g  with  G = Q1^T H_x, g = Q1^T r,
// Q1 an orthonormal basis of range(H_f).  The kernel emits the sparse rows (rec) and G|g; K2 reduces them.
#include "ovp_feat_model.h"
#include <utility>

namespace ovp {

static constexpr int NR = 64;               // rows handled by one wave (2 * OVP_MAX_MEAS)
static constexpr int TRI = NR * (NR + 1) / 2;

// compile-time loop: guarantees that register arrays are only ever indexed by constants
template <int... Is, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

__device__ __forceinline__ double rsqrt_nr(double x) {
  double y = __builtin_amdgcn_rsq(x);
  // two Newton steps: y <- y * (1.5 - 0.5 x y^2)
  double h = 0.5 * x;
  y = y * fma(-h * y, y, 1.5);
  y = y * fma(-h * y, y, 1.5);
  return y;
}

__global__ __launch_bounds__(64) void k_feat_gate(const FeatParams p) {
  const int f = blockIdx.x;
  const int lane = threadIdx.x;
  const int m = p.n_meas[f];
  const int n = 2 * m;

  __shared__ __attribute__((aligned(16))) double sJ[NR * 6];
  __shared__ __attribute__((aligned(16))) double sC[NR * 14];
  __shared__ __attribute__((aligned(16))) double sE[NR * 14];
  __shared__ __attribute__((aligned(16))) double sB[TRI + 256];

  double* sB2 = sB + 256;  // packed lower triangle of B (the first 256 doubles are scratch: P_cc, column buffers)
  const int a = lane >> 1, r = lane & 1;
  long long tstamp[8];
  tstamp[0] = __builtin_readcyclecounter();
  const bool valid = lane < n;
  const bool feat_ok = (m >= 2) && (m <= OVP_MAX_MEAS_DEV);  // UpdaterMSCKF.cpp:94-96
  const int* cidx = p.clone_idx + (size_t)f * p.max_meas;
  const int ci = cidx[valid ? a : 0];
  const int ida = p.clone_id[ci];

  // ------------------------------------------------------------------------------------------
  // Phase A: measurement model for observation a, keep row r.   UpdaterHelper.cpp:345-444
  // ------------------------------------------------------------------------------------------
  double jrow[6], crow[14], hf[3], res;
  build_bearing_row(p, f, a, r, valid, ci, jrow, crow, hf, res);
  tstamp[1] = __builtin_readcyclecounter();
  const double* P = p.P;
  const int ldp = p.ldp;
  double chi2 = 0.0;
  bool accept = false;
  double yv = 0.0, zv[3] = {0.0, 0.0, 0.0};

  if (feat_ok) {
    // ----------------------------------------------------------------------------------------
    // Phase A2: e = j P[clone(a), cal], d = c P[cal, cal], u = e + d ; publish rows to LDS
    // ----------------------------------------------------------------------------------------
    double u[14];
    {
      // 14x14 calibration block of P: one coalesced gather per wave instead of 196 serialized scalar loads
      double* sPcc = sB;  // 196 doubles, sB is not in use yet
      for (int idx = lane; idx < 196; idx += 64) {
        const int kk = idx / 14, k = idx - 14 * kk;
        const bool on = ((p.calmask >> kk) & 1) && ((p.calmask >> k) & 1);
        sPcc[idx] = on ? P[(size_t)p.calcol[kk] * ldp + p.calcol[k]] : 0.0;
      }
      double e[14];
#pragma unroll
      for (int k = 0; k < 14; ++k) {
        e[k] = 0.0;
        if ((p.calmask >> k) & 1) {
          const double* prow = P + (size_t)p.calcol[k] * ldp;
          double s = 0.0;
#pragma unroll
          for (int l = 0; l < 6; ++l) s = fma(jrow[l], prow[ida + l], s);
          e[k] = s;
        }
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 14; ++k) {
        double dsum = 0.0;
#pragma unroll
        for (int kk = 0; kk < 14; ++kk) dsum = fma(crow[kk], sPcc[kk * 14 + k], dsum);
        u[k] = e[k] + dsum;
      }
#pragma unroll
      for (int l = 0; l < 6; ++l) sJ[lane * 6 + l] = jrow[l];
#pragma unroll
      for (int k = 0; k < 14; ++k) {
        sC[lane * 14 + k] = crow[k];
        sE[lane * 14 + k] = e[k];
      }
    }
    __syncthreads();
    tstamp[2] = __builtin_readcyclecounter();

    // ----------------------------------------------------------------------------------------
    // Phase B: row `lane` of B = H_x P H_x^T + I, lower triangle, packed in LDS
    // ----------------------------------------------------------------------------------------
    {
      double pn[36];  // P[clone(b)+k][clone(a)+l] for the next b
      {
        const int idb0 = p.clone_id[cidx[0]];
#pragma unroll
        for (int k = 0; k < 6; ++k)
#pragma unroll
          for (int l = 0; l < 6; ++l) pn[6 * k + l] = P[(size_t)(idb0 + k) * ldp + ida + l];
      }
      for (int b = 0; b < m; ++b) {
        double pc[36];
#pragma unroll
        for (int q = 0; q < 36; ++q) pc[q] = pn[q];
        if (b + 1 < m) {
          const int idb1 = p.clone_id[cidx[b + 1]];
#pragma unroll
          for (int k = 0; k < 6; ++k)
#pragma unroll
            for (int l = 0; l < 6; ++l) pn[6 * k + l] = P[(size_t)(idb1 + k) * ldp + ida + l];
        }
        double t[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          double sacc = 0.0;
#pragma unroll
          for (int l = 0; l < 6; ++l) sacc = fma(jrow[l], pc[6 * k + l], sacc);
          t[k] = sacc;
        }
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
          const int col = 2 * b + rr;
          double s0 = 0.0, s1 = 0.0;
#pragma unroll
          for (int k = 0; k < 6; ++k) s0 = fma(t[k], sJ[col * 6 + k], s0);
#pragma unroll
          for (int k = 0; k < 14; ++k) s1 = fma(u[k], sC[col * 14 + k], s1);
#pragma unroll
          for (int k = 0; k < 14; ++k) s0 = fma(crow[k], sE[col * 14 + k], s0);
          if (col <= lane) sB2[tri(lane, col)] = (s0 + s1) + (col == lane ? 1.0 : 0.0);
        }
      }
    }
    __syncthreads();
    tstamp[3] = __builtin_readcyclecounter();

    // ----------------------------------------------------------------------------------------
    // Phase C: Cholesky of B with the row in registers, fused forward substitution of [r | H_f]
    // ----------------------------------------------------------------------------------------
    double arow[NR];
    static_for<NR>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      double v = (k == lane) ? 1.0 : 0.0;
      if (valid && k <= lane) v = sB2[tri(lane, k)];
      arow[k] = v;
    });
    __syncthreads();
    double rh0 = res, rh1 = hf[0], rh2 = hf[1], rh3 = hf[2];
    double* colbuf = sB;  // 2 x 64 doubles, alternating (scratch region, disjoint from the packed triangle)
    bool spd = true;
    // look-ahead: the reciprocal square root of pivot k+1 is started as soon as column k is known, so the
    // rsq/Newton chain overlaps the LDS broadcast and the trailing FMAs of step k.
    double piv = readlane_f64(arow[0], 0);
    double inv = rsqrt_nr(piv);
    static_for<NR>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      if (k < n) {  // wave-uniform
        spd = spd && (piv > 0.0);
        const double l = arow[k] * inv;  // column k of L (valid for lanes >= k)
        const double inv_k = inv;
        double* cb = colbuf + (k & 1) * NR;
        cb[lane] = l;
        if constexpr (k + 1 < NR) {
          const double lk1 = readlane_f64(l, k + 1);
          arow[k + 1] = fma(-l, lk1, arow[k + 1]);
          piv = readlane_f64(arow[k + 1], k + 1);
          inv = rsqrt_nr(piv);
        }
        // forward substitution of the 4 right-hand sides
        const double x0 = readlane_f64(rh0, k) * inv_k, x1 = readlane_f64(rh1, k) * inv_k;
        const double x2 = readlane_f64(rh2, k) * inv_k, x3 = readlane_f64(rh3, k) * inv_k;
        if (lane > k) {
          rh0 = fma(-l, x0, rh0);
          rh1 = fma(-l, x1, rh1);
          rh2 = fma(-l, x2, rh2);
          rh3 = fma(-l, x3, rh3);
        } else if (lane == k) {
          rh0 = x0;
          rh1 = x1;
          rh2 = x2;
          rh3 = x3;
        }
        // trailing update of this lane's row: a[j] -= l_ik * l_jk   (j = k+1 already done above)
        constexpr int jb0 = (k + 2) / 8;
        static_for<8 - jb0>([&](auto jbc) {
          constexpr int j0 = (jb0 + decltype(jbc)::value) * 8;
          if (j0 < n) {  // wave-uniform
            static_for<8>([&](auto jc) {
              constexpr int j = j0 + decltype(jc)::value;
              if constexpr (j > k + 1) arow[j] = fma(-l, cb[j], arow[j]);
            });
          }
        });
      }
    });
    tstamp[4] = __builtin_readcyclecounter();
    yv = valid ? rh0 : 0.0;
    zv[0] = valid ? rh1 : 0.0;
    zv[1] = valid ? rh2 : 0.0;
    zv[2] = valid ? rh3 : 0.0;

    // ----------------------------------------------------------------------------------------
    // Phase D: chi2 = y^T y - (Z^T y)^T (Z^T Z)^-1 (Z^T y)      UpdaterMSCKF.cpp:739-764
    // ----------------------------------------------------------------------------------------
    double sums[10];
    {
      double v[16];
      v[0] = yv * yv;
      v[1] = zv[0] * yv;
      v[2] = zv[1] * yv;
      v[3] = zv[2] * yv;
      v[4] = zv[0] * zv[0];
      v[5] = zv[0] * zv[1];
      v[6] = zv[0] * zv[2];
      v[7] = zv[1] * zv[1];
      v[8] = zv[1] * zv[2];
      v[9] = zv[2] * zv[2];
#pragma unroll
      for (int k = 10; k < 16; ++k) v[k] = 0.0;
      const double rsum = wave_transpose_reduce<16>(v);
#pragma unroll
      for (int k = 0; k < 10; ++k) sums[k] = readlane_f64(rsum, reduce_owner_lane<16>(k));
    }
    {
      const double l00 = sqrt(sums[4]);
      const double l10 = sums[5] / l00, l20 = sums[6] / l00;
      const double l11 = sqrt(sums[7] - l10 * l10);
      const double l21 = (sums[8] - l20 * l10) / l11;
      const double l22 = sqrt(sums[9] - l20 * l20 - l21 * l21);
      const double w0 = sums[1] / l00;
      const double w1 = (sums[2] - l10 * w0) / l11;
      const double w2 = (sums[3] - l20 * w0 - l21 * w1) / l22;
      chi2 = sums[0] - (w0 * w0 + w1 * w1 + w2 * w2);
    }
    const int dof = n - 3;
    const double thr = p.chi2_mult * p.chi2_table[dof < OVP_CHI2_TABLE ? dof : OVP_CHI2_TABLE];
    accept = spd && (chi2 <= thr);  // NaN (rank-deficient H_f) rejects
  }
  __syncthreads();
  tstamp[5] = __builtin_readcyclecounter();

  // ------------------------------------------------------------------------------------------
  // Phase E: projector rows G = Q1^T H_x, g = Q1^T r  (Q1 by CholeskyQR2 on H_f), staged in LDS
  // ------------------------------------------------------------------------------------------
  const int ldg = p.ldg;
  double* Gst = sB;  // 3 * ldg doubles (ldg <= OVP_LDG_MAX so that it fits in TRI)
  for (int idx = lane; idx < 3 * ldg; idx += 64) Gst[idx] = 0.0;
  __syncthreads();
  if (accept) {
    double q[3] = {hf[0], hf[1], hf[2]};
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
      const double g00 = readlane_f64(rsum, reduce_owner_lane<8>(0)), g01 = readlane_f64(rsum, reduce_owner_lane<8>(1));
      const double g02 = readlane_f64(rsum, reduce_owner_lane<8>(2)), g11 = readlane_f64(rsum, reduce_owner_lane<8>(3));
      const double g12 = readlane_f64(rsum, reduce_owner_lane<8>(4)), g22 = readlane_f64(rsum, reduce_owner_lane<8>(5));
      // R upper: R^T R = G
      const double r00 = sqrt(g00), r01 = g01 / r00, r02 = g02 / r00;
      const double r11 = sqrt(g11 - r01 * r01), r12 = (g12 - r01 * r02) / r11;
      const double r22 = sqrt(g22 - r02 * r02 - r12 * r12);
      const double a0 = q[0] / r00;
      const double a1 = (q[1] - a0 * r01) / r11;
      const double a2 = (q[2] - a0 * r02 - a1 * r12) / r22;
      q[0] = a0;
      q[1] = a1;
      q[2] = a2;
    }
    // calibration columns and g: 3*14 + 3 = 45 wave sums in one transposed reduction
    {
      double v[64];
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int k = 0; k < 14; ++k) v[t * 14 + k] = q[t] * crow[k];
#pragma unroll
      for (int t = 0; t < 3; ++t) v[42 + t] = q[t] * res;
#pragma unroll
      for (int k = 45; k < 64; ++k) v[k] = 0.0;
      const double rsum = wave_transpose_reduce<64>(v);
      if (lane < 42) {
        const int t = lane / 14, k = lane - 14 * t;
        if ((p.calmask >> k) & 1) Gst[t * ldg + p.calcol[k]] = rsum;
      } else if (lane < 45) {
        Gst[(lane - 42) * ldg + p.n] = rsum;
      }
    }
    // clone columns: sum of the two rows of the observation
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int l = 0; l < 6; ++l) {
        double v = q[t] * jrow[l];
        v += shfl_xor_f64(v, 1);
        if (valid && r == 0) Gst[t * ldg + ida + l] = v;
      }
  }
  __syncthreads();
  tstamp[6] = __builtin_readcyclecounter();
  {
    double* gout = p.G + (size_t)3 * f * ldg;
    for (int idx = lane; idx < 3 * ldg; idx += 64) gout[idx] = Gst[idx];
  }

  // ------------------------------------------------------------------------------------------
  // sparse rows for K2 (zero for rejected features / unobserved clones)
  // ------------------------------------------------------------------------------------------
  unsigned long long seen = 0ull;
  if (accept) {
    for (int b = 0; b < m; ++b) seen |= 1ull << cidx[b];
    if (valid) {
      double* ro = p.rec + (((size_t)ci * p.n_feats + f) * 2 + r) * OVP_REC;
#pragma unroll
      for (int l = 0; l < 6; ++l) ro[l] = jrow[l];
#pragma unroll
      for (int k = 0; k < 14; ++k) ro[6 + k] = crow[k];
      ro[20] = res;
    }
  }
  for (int cc = 0; cc < p.n_clones; ++cc) {
    if (!((seen >> cc) & 1ull)) {
      double* ro = p.rec + (((size_t)cc * p.n_feats + f) * 2) * OVP_REC;
      if (lane < 2 * OVP_REC) ro[lane] = 0.0;
    }
  }
  tstamp[7] = __builtin_readcyclecounter();
  if (p.dbg_cycles && lane == 0) {
    for (int k = 0; k < 8; ++k) p.dbg_cycles[(size_t)f * 8 + k] = tstamp[k];
  }
  if (lane == 0) {
    p.chi2[f] = chi2;
    p.accept[f] = accept ? 1 : 0;
  }
}

}  // namespace ovp

extern "C" hipError_t ovp_launch_feat_gate(const ovp::FeatParams* p, hipStream_t stream) {
  if (p->n_feats <= 0) return hipSuccess;
  hipLaunchKernelGGL(ovp::k_feat_gate, dim3(p->n_feats), dim3(64), 0, stream, *p);
  return hipGetLastError();
}
