// UpdaterSLAM::delayed_init, candidate loop on the device (update/UpdaterSLAM.cpp:204-364, state/StateHelper.cpp:398-586).
//
// The candidates of a frame are sequential - every StateHelper::initialize ends in an EKF update that moves the poses the next
// candidate's Jacobians are evaluated at - so the loop cannot be batched, but nothing in it needs the HOST: per candidate
//
//   k_dinit_rows    one workgroup: Type::update of the device pose tables with the previous candidate's correction (its
//                   commit), the candidate's bearing rows at those tables (UpdaterHelper.cpp:345-444), the orthogonal split of
//                   H_f = Q [R3; 0] (Householder instead of the Givens sweep of StateHelper.cpp:434-446: init rows, update rows
//                   and everything derived from them are invariant under the choice of the orthogonal factor), H_L^-1 = R3^-1
//   k_init_m        (k_init.hip) M = P[:, ids] [H_init; H_up]^T on many workgroups (forming it inside k_dinit_rows - one workgroup,
//                   prefetched operand - was built in round 5, was no faster and read columns a rejected predecessor had just
//                   rewritten without a fence: removed in round 6)
//   k_init_core     (k_init.hip) chi2 of the update rows against the prior (:464-475), initialize_invertible (:520-573)
//   k_init_update   (k_init.hip) EKFUpdate with the update rows (:483-485), IN PLACE
//
// The covariance grows by three columns per candidate whatever the gate says: a rejected candidate leaves an inert block (unit
// diagonal, zero cross terms - written by the next commit) that no later product reads, so every launch geometry is known to
// the host up front and the whole loop is enqueued without a synchronisation; the host removes the inert blocks afterwards
// (ovp_cov_marginalize; rejections are rare) and applies the corrections to its own copy of the state.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "k_dinit.h"
#include "ovp_feat_model.h"

namespace ovp {

typedef double double4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dinit_rot_update(double* R, const double* dth) {
  // ext JPLQuat::update on a rotation matrix: R <- R(dq) R, dq = quatnorm([dth / 2, 1])
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
}

// workgroup barrier that orders LDS traffic only: __syncthreads() carries a fence the compiler implements with s_waitcnt vmcnt(0),
// which would wait for every global load in flight - here the prefetched operand of the M product
__device__ __forceinline__ void di_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#define DI_T 1024   // threads of the one workgroup

// (16 waves = 4 per SIMD: 128 VGPRs each; without the attribute the compiler aims at 8 waves per SIMD, stops at 64 registers and
// spills the prefetch to scratch - a dispatch that needs scratch behind ones that do not costs tens of microseconds on this stack)
__global__ __launch_bounds__(DI_T) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_dinit_rows(DinitParams dp) {
  extern __shared__ double sm[];
  const int t = threadIdx.x;
  const FeatParams& p = dp.fp;
#ifdef OVP_DI_STAMPS
  long long st[12];
  int sti = 0;
#define DI_STAMP() do { if (t == 0) st[sti++] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define DI_STAMP() do { } while (0)
#endif
  DI_STAMP();
  // Everything this kernel reads from global memory is requested HERE, in one batch: a round trip behind a kernel boundary is
  // 2-3 us (the data was just written by other CUs), and the first version paid five of them one after the other (previous
  // result -> clone ids -> tables -> clone slots -> tables again -> P).  The barriers below are LDS-only (no vmcnt(0)): the
  // prefetched operand of the M product stays in flight until it is needed.
  const int l = dp.cand;
  const int m = dp.m_obs;
  const int C = p.n_clones;
  const int ncal = __popc(p.calmask & 0x3FFFu);
  const int cols = 6 * m + ncal, rows = 2 * m, W = (cols + 4) | 1;  // [H_f (3) | H_x (cols) | res], odd pitch (LDS banks)
  const int n_prev = dp.n - 3;  // dimension in front of the previous candidate
  __shared__ double tabR[9 * OVP_MAX_CLONES], tabP[3 * OVP_MAX_CLONES], tabRf[9 * OVP_MAX_CLONES], tabPf[3 * OVP_MAX_CLONES], tabC[20];
  __shared__ double dxs[OVP_LDG_CAP], pf_s[4], okf;
  __shared__ float uv_s[2 * OVP_MAX_MEAS_DEV];
  __shared__ int cid_s[OVP_MAX_CLONES], ci_s[OVP_MAX_MEAS_DEV];
  __shared__ double beta_s, Ri[9];
  // (a) previous result + tables + this candidate's inputs
  double ld0 = 0.0, ld1 = 0.0, ld2 = 0.0, ld3 = 0.0, ld4 = 0.0;
  float lf0 = 0.f;
  int li0 = 0, li1 = 0;
  if (dp.prev_res && t < dp.n && t < OVP_LDG_CAP) ld0 = dp.prev_res[4 + t];
  if (t < 12 * C) {
    ld1 = t < 9 * C ? dp.clone_R[t] : dp.clone_p[t - 9 * C];
    ld2 = t < 9 * C ? p.clone_R_fej[t] : p.clone_p_fej[t - 9 * C];
  }
  if (t < 20) ld3 = dp.cal[t];
  if (t < C) li0 = p.clone_id[t];
  if (t == 0 && dp.prev_res) ld4 = dp.prev_res[1];
  if (l >= 0) {
    if (t < m) li1 = p.clone_idx[(size_t)l * p.max_meas + t];
    if (t < 2 * m) lf0 = p.uv[(size_t)l * p.max_meas * 2 + t];
    if (t >= 64 && t < 67) ld4 = p.p_FinG[3 * l + (t - 64)];
  }
  DI_STAMP();
  DI_STAMP();
  // (c) into LDS
  if (t < OVP_LDG_CAP) dxs[t] = ld0;
  if (t < 9 * C) tabR[t] = ld1, tabRf[t] = ld2;
  else if (t < 12 * C) tabP[t - 9 * C] = ld1, tabPf[t - 9 * C] = ld2;
  if (t < 20) tabC[t] = ld3;
  if (t < C) cid_s[t] = li0;
  if (t == 0) okf = ld4;
  if (l >= 0) {
    if (t < m) ci_s[t] = li1;
    if (t < 2 * m) uv_s[t] = lf0;
    if (t >= 64 && t < 67) pf_s[t - 64] = ld4;
  }
  di_lds_barrier();
  DI_STAMP();
  // ---- commit of the previous candidate (StateHelper.cpp:188-194 Type::update; the host repeats it on its own copy): the tables
  // in LDS are what the rows below are built at, the global copies what later kernels read ----
  if (dp.prev_res) {
    const bool ok = okf > 0.5;
    if (ok) {
      if (t < C) {
        const int id = cid_s[t];
        dinit_rot_update(tabR + 9 * t, dxs + id);
        for (int k = 0; k < 3; ++k) tabP[3 * t + k] += dxs[id + 3 + k];
        for (int k = 0; k < 9; ++k) dp.clone_R[9 * t + k] = tabR[9 * t + k];
        for (int k = 0; k < 3; ++k) dp.clone_p[3 * t + k] = tabP[3 * t + k];
      } else if (t == 64) {
        if (p.calmask & 0x3Fu) {
          dinit_rot_update(tabC, dxs + p.calcol[0]);
          for (int k = 0; k < 3; ++k) tabC[9 + k] += dxs[p.calcol[3] + k];
        }
        if (p.calmask & (0xFFu << 6))
          for (int k = 0; k < 8; ++k) tabC[12 + k] += dxs[p.calcol[6] + k];
        for (int k = 0; k < 20; ++k) dp.cal[k] = tabC[k];
      }
    } else {
      // rejected: its three columns stay as an inert block (nobody reads it; the host removes it after the loop)
      double* P = dp.P;
      for (int e = t; e < 3 * dp.n; e += DI_T) {
        const int k = e / dp.n, r = e - k * dp.n;
        const double v = (r == n_prev + k) ? 1.0 : 0.0;
        P[(size_t)r * p.ldp + n_prev + k] = v;
        P[(size_t)(n_prev + k) * p.ldp + r] = v;
      }
    }
  }
  if (l < 0) return;  // commit only (behind the last candidate)
  double* A = sm;                 // [rows][W] row-major
  double* v = A + (size_t)rows * W;  // [rows] Householder vector
  int* ids_s = (int*)(v + rows + 8);
  DI_STAMP();
  for (int e = t; e < rows * W; e += DI_T) A[e] = 0.0;
  for (int e = t; e < cols; e += DI_T) ids_s[e] = dp.idv[e];
  di_lds_barrier();
  DI_STAMP();
  if (t < rows) {
    const int a = t >> 1, r = t & 1;
    // the measurement model on the LDS copies: tables as the commit above left them, this candidate's inputs as feature 0
    FeatParams q = p;
    q.clone_R = tabR;
    q.clone_p = tabP;
    q.clone_R_fej = tabRf;
    q.clone_p_fej = tabPf;
    q.cal = tabC;
    q.uv = uv_s;
    q.p_FinG = pf_s;
    double jrow[6], crow[14], hf[3], res;
    build_bearing_row(q, 0, a, r, true, ci_s[a], jrow, crow, hf, res);  // (first estimate of the new landmark = its value, :240-246)
    double* h = A + (size_t)t * W;
    h[0] = hf[0], h[1] = hf[1], h[2] = hf[2];
#pragma unroll
    for (int k = 0; k < 6; ++k) h[3 + 6 * a + k] = jrow[k];
#pragma unroll
    for (int k = 0; k < 14; ++k)
      if ((p.calmask >> k) & 1) h[3 + 6 * m + __popc(p.calmask & ((1u << k) - 1u))] = crow[k];
    h[3 + cols] = res;
  }
  di_lds_barrier();
  DI_STAMP();
  // ---- H_f = Q [R3; 0]: three reflectors applied to [H_f | H_x | res] ----
  for (int j = 0; j < 3; ++j) {
    if (t < 64) {  // wave 0: |x|^2 of column j below the diagonal by a wave reduction
      double part = 0.0;
      for (int i = j + t; i < rows; i += 64) part = fma(A[(size_t)i * W + j], A[(size_t)i * W + j], part);
      part += xor_lane_f64<1>(part);   // (DPP + row swaps: plain VALU; wave_sum's ds_bpermute round trips were 1.5 us per reflector)
      part += xor_lane_f64<2>(part);
      part += xor_lane_f64<4>(part);
      part += xor_lane_f64<8>(part);
      const double nn = rows_sum_f64(part);
      if (t == 0) {
        const double x0 = A[(size_t)j * W + j];
        const double alpha = x0 >= 0.0 ? -sqrt(nn) : sqrt(nn);
        const double v0 = x0 - alpha;
        const double vv = nn - x0 * x0 + v0 * v0;
        beta_s = vv > 0.0 ? 2.0 / vv : 0.0;
        v[j] = v0;
      }
    }
    for (int i = j + 1 + t; i < rows; i += DI_T) v[i] = A[(size_t)i * W + j];
    di_lds_barrier();
    // eight lanes per column, each an eighth of the rows; the partial dot products meet inside the group of eight (DPP); columns
    // beyond 128 take a second pass
    for (int cb = j; cb < W; cb += DI_T / 8) {
      const int c = cb + (t >> 3), part = t & 7;
      double s = 0.0;
      if (c < W)
        for (int i = j + part; i < rows; i += 8) s = fma(v[i], A[(size_t)i * W + c], s);
      s += xor_lane_f64<1>(s);
      s += xor_lane_f64<2>(s);
      s += xor_lane_f64<4>(s);
      s *= beta_s;
      if (c < W)
        for (int i = j + part; i < rows; i += 8) A[(size_t)i * W + c] = fma(-s, v[i], A[(size_t)i * W + c]);
    }
    di_lds_barrier();
  }
  DI_STAMP();
  if (t == 0) {
    // R3 upper triangular (rows 0..2 of the H_f columns); H_L^-1 = R3^-1
    const double r00 = A[0], r01 = A[1], r02 = A[2], r11 = A[W + 1], r12 = A[W + 2], r22 = A[2 * W + 2];
    const double i00 = 1.0 / r00, i11 = 1.0 / r11, i22 = 1.0 / r22;
    Ri[0] = i00, Ri[1] = -r01 * i00 * i11, Ri[2] = (r01 * r12 - r02 * r11) * i00 * i11 * i22;
    Ri[3] = 0.0, Ri[4] = i11, Ri[5] = -r12 * i11 * i22;
    Ri[6] = 0.0, Ri[7] = 0.0, Ri[8] = i22;
    // H_Linv * res_init: what the host adds to the new landmark's value (StateHelper.cpp:577)
    const double q0 = A[3 + cols], q1 = A[W + 3 + cols], q2 = A[2 * W + 3 + cols];
    dp.res[4 + dp.n_max + 0] = Ri[0] * q0 + Ri[1] * q1 + Ri[2] * q2;
    dp.res[4 + dp.n_max + 1] = Ri[4] * q1 + Ri[5] * q2;
    dp.res[4 + dp.n_max + 2] = Ri[8] * q2;
    for (int i = 0; i < 9; ++i) dp.Hinv[i] = Ri[i];           // row-major [3][3]
    for (int i = 0; i < 9; ++i) dp.Rk[i] = (i % 4 == 0) ? 1.0 : 0.0;  // R_init = I (:304 R = identity)
  }
  // ---- the stacked transposed system H_all^T [cols][rows] (init rows 0..2, update rows behind) and the update residual ----
  for (int e = t; e < cols * rows; e += DI_T) {
    const int a = e / rows, i = e - a * rows;
    dp.Ht[e] = A[(size_t)i * W + 3 + a];
  }
  for (int i = 3 + t; i < rows; i += DI_T) dp.resid[i - 3] = A[(size_t)i * W + 3 + cols];
}

}  // namespace ovp

extern "C" {
size_t ovp_dinit_rows_lds(int m_obs, int ncal) {
  const int rows = 2 * m_obs, cols = 6 * m_obs + ncal, W = (cols + 4) | 1;
  return sizeof(double) * ((size_t)rows * W + rows + 8 + (size_t)(cols + 2) / 2 + 2);
}

hipError_t ovp_launch_dinit_rows(const ovp::DinitParams* dp, size_t lds, hipStream_t stream) {
  static unsigned long long attr_mask = 0;  // per device (ovp_kernels.h)
  if (ovp_lds_attr_needed(&attr_mask)) {
    (void)hipFuncSetAttribute((const void*)ovp::k_dinit_rows, hipFuncAttributeMaxDynamicSharedMemorySize, OVP_DINIT_DYN_LDS);
    (void)hipGetLastError();  // (a kernel with static LDS refuses the full 160 KB: harmless, a real shortage fails the launch itself)
    ovp_lds_attr_done(&attr_mask);
  }
  hipLaunchKernelGGL(ovp::k_dinit_rows, dim3(1), dim3(DI_T), lds, stream, *dp);
  return hipGetLastError();
}
}
