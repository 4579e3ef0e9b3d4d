// StateHelper::initialize downstream of its Givens split (state/StateHelper.cpp:448-487) as three kernels: the Mahalanobis gate
// on the prior (:464-475), StateHelper::initialize_invertible (:489-586) and the EKF update with the remaining rows in the
// reference's own S-form (StateHelper::EKFUpdate, :121-202).  The update rows of one landmark are few (2 m - 3): S = H P H^T + R
// is a small LDS matrix, and P+ = P - W W^T with W = P H^T L^-T is a rank-rup downdate every tile of P takes independently.
//
// Everything here is latency-bound (a few hundred kFLOP): the kernels are laid out so that no thread runs a loop of dependent
// global loads - gathers are staged into LDS by all threads at once, the K-loops then read LDS or unit-stride global memory.
//
//   H_all = [H_init ; H_up]   (m = k + rup rows), handed over transposed: Ht [cols][m]
//   k_init_m       M_all = P[:, ids] H_all^T                       n x m, many workgroups
//   k_init_core    S = H_up M_up[ids] + r I = L L^T, chi2 = |L^-1 res|^2, L^-1;  the new rows / columns of P   one workgroup
//   k_init_update  P+ = P - W W^T, dx = W y  with  W = M_up L^-T,  y = L^-1 res    (n + k)^2 / 256 workgroups
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ovp_dev.h"
#include "ovp_kernels.h"

namespace ovp {

#define IM_ROWS 8
#define IC_MAX_COLS 704  // = OVP_LDG_CAP: columns of one dense block
#define IC_MAXE 13  // elements of [S | res | I] a thread of k_init_core owns: ceil(80 * 161 / 1024)
// grid = ceil(n / 8), 256 threads = 8 rows x 32 column lanes; dynamic LDS: 8 x cols doubles (+ cols x m for H^T, hs_in_lds)
__global__ __launch_bounds__(256) void k_init_m(const double* __restrict__ P, int ldp, int n, const int* __restrict__ ids, int cols,
                                                 const double* __restrict__ Ht, int m, double* __restrict__ Mall, int hs_in_lds) {
  extern __shared__ double prow[];  // [8][cols]
  double* Hl = prow + IM_ROWS * cols;
  const int t = threadIdx.x;
  const int i0 = blockIdx.x * IM_ROWS;
  for (int e = t; e < IM_ROWS * cols; e += 256) {
    const int r = e / cols, a = e - r * cols;
    const int i = i0 + r;
    prow[e] = i < n ? P[(size_t)i * ldp + ids[a]] : 0.0;
  }
  if (hs_in_lds)
    for (int e = t; e < cols * m; e += 256) Hl[e] = Ht[e];
  __syncthreads();
  const double* Hs = hs_in_lds ? Hl : Ht;
  const int r = t >> 5, jl = t & 31;
  const int i = i0 + r;
  if (i >= n) return;
  const double* pr = prow + r * cols;
  for (int j = jl; j < m; j += 32) {
    double s0 = 0.0, s1 = 0.0;
    int a = 0;
#pragma unroll 4
    for (; a + 1 < cols; a += 2) {
      s0 = fma(pr[a], Hs[(size_t)a * m + j], s0);
      s1 = fma(pr[a + 1], Hs[(size_t)(a + 1) * m + j], s1);
    }
    if (a < cols) s0 = fma(pr[a], Hs[(size_t)a * m + j], s0);
    Mall[(size_t)i * m + j] = s0 + s1;
  }
}

template <int NE>
__device__ __forceinline__ void ic_eliminate(double* Wm, int ldw, int rup, int W, int t, int* bad) {
  const int tot = rup * W;
  double v[NE];
  int ei[NE], ej[NE];
#pragma unroll
  for (int q = 0; q < NE; ++q) {
    const int e = t + 1024 * q;
    ei[q] = e < tot ? e / W : 0;      // (an element of row 0 is never touched: i > c fails for every c)
    ej[q] = e < tot ? e - ei[q] * W : 0;
    v[q] = Wm[ei[q] * ldw + ej[q]];
  }
  for (int c = 0; c < rup; ++c) {
    const double* rc = Wm + c * ldw;
    const double piv = rc[c];
    if (t == 0 && !(piv > 0.0)) *bad = 1;
    const double ip = 1.0 / piv;
#pragma unroll
    for (int q = 0; q < NE; ++q) {
      if (ei[q] > c && ej[q] > c) {
        v[q] = fma(-(rc[ei[q]] * ip), rc[ej[q]], v[q]);
        if (ei[q] == c + 1) Wm[(c + 1) * ldw + ej[q]] = v[q];
      }
    }
    __syncthreads();
  }
}

// One workgroup of 1024.  Dynamic LDS: Mg [cols][m] | Wm [rup][2 rup + 2] | Hs [cols][m] (only when it fits, hs_in_lds).
//   res[0] = chi2, res[1] = 1 accept / 0 reject (a non-positive pivot of S rejects), res[2] = 0 (the update's negative-diagonal
//   mark).  With rup == 0 there is no gate: res = {0, 1, 0}.
// Writes the k new rows / columns of P (harmless when the gate says no: the dimension then stays n), rows n .. n + k of M_up,
// Linv [rup][rup] row-major and y.
// The factorization is Gaussian elimination of [S | res | I] without pivoting, one barrier per column: row c of the reduced
// matrix is final after step c - 1, and divided by the square root of its pivot it is [L^T row c | y_c | L^-1 row c].
__global__ __launch_bounds__(1024) void k_init_core(double* __restrict__ P, int ldp, int n, const int* __restrict__ ids, int cols,
                                                     const double* __restrict__ Ht, int k, int rup, double* __restrict__ Mall,
                                                     const double* __restrict__ Hinv, const double* __restrict__ Rk,
                                                     const double* __restrict__ resid, double r_iso, double thr,
                                                     double* __restrict__ Linv, double* __restrict__ y, double* __restrict__ res,
                                                     int hs_in_lds) {
  extern __shared__ double sm[];
  const int t = threadIdx.x;
#ifdef OVP_IC_STAMPS
  long long st[10];
  int sti = 0;
#define IC_STAMP() do { if (t == 0) st[sti++] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define IC_STAMP() do { } while (0)
#endif
  IC_STAMP();
  const int m = k + rup, W = 2 * rup + 1, ldw = W + 1;
  double* Mg = sm;                       // [cols][m]: row ids[a] of M_all
  double* Wm = Mg + (size_t)cols * m;    // [rup][2 rup + 2]: S | res | I
  double* Hl = Wm + (size_t)rup * ldw;   // [cols][m] copy of Ht
  __shared__ double Minit[36], PLL[36], Hi[36], X[6 * 80];
  __shared__ int bad;
  if (t == 0) bad = 0;
  if (t < k * k) Hi[t] = Hinv[t];
  __shared__ int ids_s[IC_MAX_COLS];  // (the gather below then has ONE global load per element, not a dependent pair)
  // the init columns of M that the new rows / columns of P are made of (step further down) are requested now: by then the round trip
  // (2-3 us behind a kernel boundary) is over instead of starting
  double pm[2][6];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int e = t + 1024 * q, r = k > 0 ? e / k : 0;
#pragma unroll
    for (int a = 0; a < 6; ++a) pm[q][a] = (k > 0 && e < n * k && a < k) ? Mall[(size_t)r * m + a] : 0.0;
  }
  for (int e = t; e < cols; e += 1024) ids_s[e] = ids[e];
  __syncthreads();
  IC_STAMP();
#pragma unroll 4
  for (int e = t; e < cols * m; e += 1024) {
    const int a = e / m, j = e - a * m;
    Mg[e] = Mall[(size_t)ids_s[a] * m + j];
    if (hs_in_lds) Hl[e] = Ht[e];
  }
  for (int e = t; e < rup * (rup + 1); e += 1024) {  // [res | I]
    const int i = e / (rup + 1), q = e - i * (rup + 1);
    Wm[i * ldw + rup + q] = q == 0 ? resid[i] : (q - 1 == i ? 1.0 : 0.0);
  }
  __syncthreads();
  IC_STAMP();
  const double* Hs = hs_in_lds ? Hl : Ht;
  // S = H_up M_up[ids] + r I ;  Minit = H_init M_init[ids] + R (upper triangle mirrored, selfadjointView<Upper>) ;
  // X = H_init M_up[ids]  (k x rup)
  // many update rows (a frame's landmark re-observations): S in 2 x 2 register blocks - four LDS reads per four FMAs instead of
  // eight (the element-per-thread loop below is bound by the LDS bandwidth at 50 rows x 95 columns: 15 us)
  const bool blocked = rup >= 24;
  if (blocked) {
    const int rb = (rup + 1) >> 1;
    for (int e = t; e < rb * rb; e += 1024) {
      const int i2 = e / rb, j2 = e - i2 * rb;
      const int i0 = 2 * i2, j0 = 2 * j2;
      const int i1 = i0 + 1 < rup ? i0 + 1 : i0, j1 = j0 + 1 < rup ? j0 + 1 : j0;
      double s00 = 0.0, s01 = 0.0, s10 = 0.0, s11 = 0.0;
#pragma unroll 4
      for (int a = 0; a < cols; ++a) {
        const double h0 = Hs[(size_t)a * m + k + i0], h1 = Hs[(size_t)a * m + k + i1];
        const double m0 = Mg[a * m + k + j0], m1 = Mg[a * m + k + j1];
        s00 = fma(h0, m0, s00);
        s01 = fma(h0, m1, s01);
        s10 = fma(h1, m0, s10);
        s11 = fma(h1, m1, s11);
      }
      Wm[i0 * ldw + j0] = s00 + (i0 == j0 ? r_iso : 0.0);
      if (j1 != j0) Wm[i0 * ldw + j1] = s01 + (i0 == j1 ? r_iso : 0.0);
      if (i1 != i0) Wm[i1 * ldw + j0] = s10 + (i1 == j0 ? r_iso : 0.0);
      if (i1 != i0 && j1 != j0) Wm[i1 * ldw + j1] = s11 + (i1 == j1 ? r_iso : 0.0);
    }
  }
  for (int e = t + (blocked ? rup * rup : 0); e < rup * rup + k * k + k * rup; e += 1024) {
    int hi, mj;      // row of H_all, column of M_all
    double s;
    double* dst;
    if (e < rup * rup) {
      const int i = e / rup, j = e - i * rup;
      hi = k + i, mj = k + j, s = (i == j) ? r_iso : 0.0, dst = Wm + i * ldw + j;
    } else if (e < rup * rup + k * k) {
      const int q = e - rup * rup, i = q / k, j = q - i * k;
      const int ii = i <= j ? i : j, jj = i <= j ? j : i;
      hi = ii, mj = jj, s = Rk[ii * k + jj], dst = Minit + q;
    } else {
      const int q = e - rup * rup - k * k, i = q / rup, j = q - i * rup;
      hi = i, mj = k + j, s = 0.0, dst = X + q;
    }
    double s1 = 0.0;
    int a = 0;
#pragma unroll 4
    for (; a + 1 < cols; a += 2) {
      s = fma(Hs[(size_t)a * m + hi], Mg[a * m + mj], s);
      s1 = fma(Hs[(size_t)(a + 1) * m + hi], Mg[(a + 1) * m + mj], s1);
    }
    if (a < cols) s = fma(Hs[(size_t)a * m + hi], Mg[a * m + mj], s);
    *dst = s + s1;
  }
  __syncthreads();
  IC_STAMP();
  // P_LL = Hinv Minit Hinv^T ; P[0:n, n:n+k] = -M_init Hinv^T and its transpose ; rows n .. n + k of M_up = -Hinv X
  if (t < k * k) {
    const int i = t / k, j = t - i * k;
    double s = 0.0;
    for (int a = 0; a < k; ++a)
      for (int b = 0; b < k; ++b) s = fma(Hi[i * k + a] * Minit[a * k + b], Hi[j * k + b], s);
    P[(size_t)(n + i) * ldp + n + j] = s;
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int e = t + 1024 * q;
    if (k > 0 && e < n * k) {
      const int r = e / k, j = e - r * k;
      double s = 0.0;
#pragma unroll
      for (int a = 0; a < 6; ++a)
        if (a < k) s = fma(pm[q][a], Hi[j * k + a], s);
      P[(size_t)r * ldp + n + j] = -s;
      P[(size_t)(n + j) * ldp + r] = -s;
    }
  }
  for (int e = t + 2048; e < n * k; e += 1024) {
    const int r = e / k, j = e - r * k;
    double s = 0.0;
    for (int a = 0; a < k; ++a) s = fma(Mall[(size_t)r * m + a], Hi[j * k + a], s);
    P[(size_t)r * ldp + n + j] = -s;
    P[(size_t)(n + j) * ldp + r] = -s;
  }
  for (int e = t; e < k * rup; e += 1024) {
    const int i = e / rup, j = e - i * rup;
    double s = 0.0;
    for (int b = 0; b < k; ++b) s = fma(Hi[i * k + b], X[b * rup + j], s);
    Mall[(size_t)(n + i) * m + k + j] = -s;
  }
  if (rup == 0) {
    if (t == 0) res[0] = 0.0, res[1] = 1.0, res[2] = 0.0;
    return;
  }
  IC_STAMP();
  // elimination: Wm[i][j] -= Wm[i][c] Wm[c][j] / Wm[c][c] for i, j > c, with the matrix in REGISTERS (a thread owns the elements
  // t, t + 1024, ... - at most 13 of the 80 x 161) and only the pivot row in LDS: row c + 1 is final after step c and its owners
  // publish it, everybody reads its two operands from the published row c - the multiplier S[i][c] is taken as S[c][i], which the
  // elimination of a symmetric S keeps equal up to rounding.  One barrier, two LDS reads and one FMA per element and step (the
  // first version walked the LDS copy with a division per element: 1.2 us per step at 50 rows, 87 us per SLAM update).
  const int ne = (rup * W + 1023) >> 10;  // elements per thread: the loop is instantiated per count (a fixed 13-fold unrolled one made
                                          // the compiler issue all 26 LDS reads of a step speculatively, whatever the guards said)
  if (ne <= 1) ic_eliminate<1>(Wm, ldw, rup, W, t, &bad);
  else if (ne <= 2) ic_eliminate<2>(Wm, ldw, rup, W, t, &bad);
  else if (ne <= 3) ic_eliminate<3>(Wm, ldw, rup, W, t, &bad);
  else if (ne <= 5) ic_eliminate<5>(Wm, ldw, rup, W, t, &bad);
  else if (ne <= 8) ic_eliminate<8>(Wm, ldw, rup, W, t, &bad);
  else ic_eliminate<IC_MAXE>(Wm, ldw, rup, W, t, &bad);
  IC_STAMP();
  // rows scaled by 1 / sqrt(pivot): y and L^-1
  for (int e = t; e < rup * (rup + 1); e += 1024) {
    const int i = e / (rup + 1), q = e - i * (rup + 1);
    const double piv = Wm[i * ldw + i];
    const double v = Wm[i * ldw + rup + q] / sqrt(piv > 0.0 ? piv : 1.0);
    if (q == 0) y[i] = v;
    else Linv[i * rup + q - 1] = q - 1 <= i ? v : 0.0;
    if (q == 0) Wm[i * ldw + rup] = v;
  }
  __syncthreads();
  if (t < 64) {  // |y|^2 by wave 0 (rup <= 80: two entries per lane), DPP + row swaps
    double part = 0.0;
    for (int j = t; j < rup; j += 64) part = fma(Wm[j * ldw + rup], Wm[j * ldw + rup], part);
    part += xor_lane_f64<1>(part);
    part += xor_lane_f64<2>(part);
    part += xor_lane_f64<4>(part);
    part += xor_lane_f64<8>(part);
    const double chi2 = rows_sum_f64(part);
    if (t == 0) {
    res[0] = chi2;
    res[1] = (!bad && !(chi2 > thr)) ? 1.0 : 0.0;
    res[2] = 0.0;
    }
  }
#ifdef OVP_IC_STAMPS
  IC_STAMP();
  if (t == 0)
    printf("[k_init_core k=%d rup=%d cols=%d n=%d] ids %lld | gather %lld | S %lld | new rows %lld | eliminate %lld | finalize %lld\n", k, rup, cols, n,
           st[1] - st[0], st[2] - st[1], st[3] - st[2], st[4] - st[3], st[5] - st[4], st[6] - st[5]);
#endif
}

#define IU_T 16
// Grid (T, T), T = ceil(n2 / 16); 256 threads.  Dynamic LDS: Li [rup][rup] | Mi, Mj [16][rup] | Wi, Wj [16][rup + 1].
// Pdst[tile] = Psrc[tile] - W_I W_J^T; the tiles of block column 0 also leave dx = W y.  Nothing is written when the gate said
// no (res[1] == 0): the caller then keeps Psrc.
__global__ __launch_bounds__(256) void k_init_update(const double* Psrc, double* Pdst /* may be Psrc: an element is read and written by its own thread only */, int ldp, int n2,
                                                      const double* __restrict__ Mall, int m, int k, int rup,
                                                      const double* __restrict__ Linv, const double* __restrict__ y,
                                                      double* __restrict__ res, double* __restrict__ dx) {
  if (res[1] == 0.0) return;
  extern __shared__ double sm[];
  double* Li = sm;                               // [rup][rup]
  double* Mi = Li + (size_t)rup * rup;           // [16][rup]
  double* Mj = Mi + IU_T * rup;
  double* Wi = Mj + IU_T * rup;                  // [16][rup + 1]
  double* Wj = Wi + IU_T * (rup + 1);
  const int t = threadIdx.x;
  const int I = blockIdx.y, J = blockIdx.x;
  for (int e = t; e < rup * rup; e += 256) Li[e] = Linv[e];
  for (int e = t; e < 2 * IU_T * rup; e += 256) {
    const int half = e / (IU_T * rup), q = e - half * IU_T * rup;
    const int r = q / rup, j = q - r * rup;
    const int i = (half ? J : I) * IU_T + r;
    (half ? Mj : Mi)[q] = i < n2 ? Mall[(size_t)i * m + k + j] : 0.0;
  }
  __syncthreads();
  for (int e = t; e < 2 * IU_T * rup; e += 256) {
    const int half = e / (IU_T * rup), q = e - half * IU_T * rup;
    const int r = q / rup, j = q - r * rup;
    const double* mr = (half ? Mj : Mi) + r * rup;
    const double* lj = Li + j * rup;
    double s = 0.0;
    for (int c = 0; c <= j; ++c) s = fma(mr[c], lj[c], s);
    (half ? Wj : Wi)[r * (rup + 1) + j] = s;
  }
  __syncthreads();
  const int r = t >> 4, c = t & 15;
  const int i = I * IU_T + r, j = J * IU_T + c;
  if (i < n2 && j < n2) {
    const double* wi = Wi + r * (rup + 1);
    const double* wj = Wj + c * (rup + 1);
    double s = 0.0;
    for (int q = 0; q < rup; ++q) s = fma(wi[q], wj[q], s);
    const double v = Psrc[(size_t)i * ldp + j] - s;
    Pdst[(size_t)i * ldp + j] = v;
    if (i == j && v < 0.0) res[2] = 1.0;
  }
  if (J == 0 && t < IU_T && I * IU_T + t < n2) {
    const double* wi = Wi + t * (rup + 1);
    double s = 0.0;
    for (int q = 0; q < rup; ++q) s = fma(wi[q], y[q], s);
    dx[I * IU_T + t] = s;
  }
}

}  // namespace ovp

extern "C" {
// LDS bytes of k_init_core without / with its copy of H^T; the caller refuses problems above ovp_init_max_lds()
size_t ovp_init_core_lds(int k, int rup, int cols) {
  return sizeof(double) * ((size_t)cols * (k + rup) + (size_t)rup * (2 * rup + 2));
}
size_t ovp_init_max_lds() { return 152 * 1024; }
int ovp_init_max_rows() { return 80; }

hipError_t ovp_launch_init_m(const double* P, int ldp, int n, const int* ids, int cols, const double* Ht, int m, double* Mall,
                             hipStream_t stream) {
  size_t lds = sizeof(double) * IM_ROWS * cols;
  const size_t hs = sizeof(double) * (size_t)cols * m;
  const int hs_in_lds = lds + hs <= 60 * 1024;  // keeps several workgroups per CU resident
  if (hs_in_lds) lds += hs;
  hipLaunchKernelGGL(ovp::k_init_m, dim3((n + IM_ROWS - 1) / IM_ROWS), dim3(256), lds, stream, P, ldp, n, ids, cols, Ht, m, Mall,
                     hs_in_lds);
  return hipGetLastError();
}

hipError_t ovp_launch_init_core(double* P, int ldp, int n, const int* ids, int cols, const double* Ht, int k, int rup, double* Mall,
                                const double* Hinv, const double* Rk, const double* resid, double r_iso, double thr, double* Linv,
                                double* y, double* res, hipStream_t stream) {
  static unsigned long long attr_mask = 0;  // per device (ovp_kernels.h)
  if (ovp_lds_attr_needed(&attr_mask)) {
    (void)hipFuncSetAttribute((const void*)ovp::k_init_core, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ovp_init_max_lds());
    (void)hipFuncSetAttribute((const void*)ovp::k_init_update, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ovp_init_max_lds());
    (void)hipFuncSetAttribute((const void*)ovp::k_init_m, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ovp_init_max_lds());
    (void)hipGetLastError();  // (a kernel with static LDS refuses the full 160 KB: harmless, a real shortage fails the launch itself)
    ovp_lds_attr_done(&attr_mask);
  }
  size_t lds = ovp_init_core_lds(k, rup, cols);
  const size_t hs = sizeof(double) * (size_t)cols * (k + rup);
  const int hs_in_lds = lds + hs <= ovp_init_max_lds();
  if (hs_in_lds) lds += hs;
  hipLaunchKernelGGL(ovp::k_init_core, dim3(1), dim3(1024), lds, stream, P, ldp, n, ids, cols, Ht, k, rup, Mall, Hinv, Rk, resid, r_iso,
                     thr, Linv, y, res, hs_in_lds);
  return hipGetLastError();
}

hipError_t ovp_launch_init_update(const double* Psrc, double* Pdst, int ldp, int n2, const double* Mall, int m, int k, int rup,
                                  const double* Linv, const double* y, double* res, double* dx, hipStream_t stream) {
  const int T = (n2 + IU_T - 1) / IU_T;
  const size_t lds = sizeof(double) * ((size_t)rup * rup + 2 * IU_T * rup + 2 * IU_T * (rup + 1));
  hipLaunchKernelGGL(ovp::k_init_update, dim3(T, T), dim3(256), lds, stream, Psrc, Pdst, ldp, n2, Mall, m, k, rup, Linv, y, res, dx);
  return hipGetLastError();
}
}
