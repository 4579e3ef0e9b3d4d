// K3 fast path: register-resident tile Cholesky + blocked forward substitution on v_mfma_f64_16x16x4_f64.
//
// A 200-300 dimensional factorization is latency bound, not FLOP bound: the 3 MFLOP of chol(210) are 10 us of one
// CU's f64 rate, but every column step of a textbook kernel costs a barrier and a memory round trip.  Here the
// whole lower triangle lives in the register file of ONE workgroup (8 waves x <=22 tiles of 16x16 f64 in MFMA
// accumulator layout = up to 360 KB of the CU's 512 KB), panels are exchanged through LDS, and every step that
// is not the 16x16 diagonal factorization is an MFMA:
//   step k:  (a) owner publishes tile (k,k)             -> LDS  (already during the trailing phase of step k-1)
//            (b) wave 0: lane-per-row Cholesky of the 16x16 block in registers (v_readlane broadcasts, no
//                barriers) and its inverse  X = L_kk^-1 ; publishes L_kk and W = X^T
//            (c) panel tiles (i,k) <- tile * W           (4 MFMA each), published to LDS
//            (d) trailing tiles (i,j) -= L_ik L_jk^T      (4 MFMA each)
// Two workgroup barriers per 16 columns instead of ~3 per column.
// f64 MFMA operand layout (cdna_hip_programming.md §3): A[i = lane&15][k = lane>>4], B[k = lane>>4][j = lane&15],
// C/D reg v: row = (lane>>4) + 4 v, col = lane&15.
#include "k_tile_body.h"

namespace ovp {

template <int MAXSLOT>
__global__ __launch_bounds__(TC_WAVES * 64) void k_tilechol(const double* __restrict__ A, double* __restrict__ L,
                                                           double* __restrict__ Dinv, double* __restrict__ Lpack, int n,
                                                           int ld, int* __restrict__ flag, int add_identity,
                                                           int dbg_skip, const int* __restrict__ cond) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  // cond = {have, want} on the device: the factor this launch would produce is already there when the two agree (the last
  // accepted plane of the plane loop left Lpack / Dinv behind itself, k_chol2) - decided on the device, no host round trip
  if (cond && cond[0] != 0 && cond[0] == cond[1]) return;
  tilechol_body<MAXSLOT>(A, L, Dinv, Lpack, n, ld, flag, add_identity, dbg_skip, lds);
}

// ------------------------------------------------------------------------------------------------
// V = Lt^-1 * M  (M = L^T in the plain update) by blocked forward substitution, right-looking; one workgroup (4 waves)
// per 16-column slab of V.  Wave w owns the tile rows i = w, w+4, ... of the slab and keeps their accumulators
//   acc_i = M_i - sum_{k<i, k done} Lt_ik V_k
// in MFMA C layout, which is also the B-operand layout, so the diagonal solve  V_k = Dinv_k acc_k  (Dinv_k = Lt_kk^-1 from
// k_tilechol) needs no data movement at all.  Per step: the owner of row k multiplies, publishes V_k to LDS (double
// buffered), ONE barrier, every wave updates its rows i > k - row k+1 first, whose owner then carries on with the next
// diagonal solve while the others finish.  Lt comes tile-packed (Ltp, column-major tiles = coalesced A operands) and is
// prefetched one step ahead; nothing on the dependent chain touches global memory.
// ------------------------------------------------------------------------------------------------
static constexpr int FW_ROWS = 5;  // tile rows per wave: nt <= 20

__global__ __launch_bounds__(256) void k_fwdsub(const double* __restrict__ Ltp, const double* __restrict__ Dinv,
                                                 const double* __restrict__ Lmat, double* __restrict__ V, int n,
                                                 int ld, int dense, int n_lead) {
  // n_lead < n: Lt = blockdiag(Lt_lead, I) - the packed factor and the inverted diagonal blocks belong to the leading n_lead
  // columns (laid out for them), the rows behind are copied
  __shared__ __attribute__((aligned(16))) double vt[2][TSZ];
  const int nt = (n + 15) >> 4;
  const int ntl = (n_lead > 0 && n_lead < n) ? ((n_lead + 15) >> 4) : nt;
  const int cblk = blockIdx.x;  // column tile of V
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane >> 4, lc = lane & 15;
  auto tile_index = [&](int i, int k) { return k * ntl - (k * (k - 1)) / 2 + (i - k); };

  // accumulators = right-hand side tiles: element [row][col] of tile i is M[16 i + row][16 cblk + col];
  // for M = L^T (dense == 0) that is L[16 cblk + col][16 i + row], zero for i > cblk
  // A operands (tiles of Lt) are fetched TWO steps ahead into a ring of three register sets (round 6): with one step of lead the
  // copy "next -> current" at the end of a step waited for loads issued ~1000 cycles earlier - a global round trip (~3000 cycles
  // from another XCD's L2) stalled every one of the nt steps; the ring needs no copies, so a load is waited for two steps after
  // its issue
  double4_t acc[FW_ROWS], ring[3][FW_ROWS];
  sfor<FW_ROWS>([&](auto rc) {
    constexpr int r = decltype(rc)::value;
    const int i = wave + 4 * r;
    double4_t t = {0.0, 0.0, 0.0, 0.0};
    // dense: 0 = M = L^T with L lower triangular (tiles below the diagonal of M are zero), 1 = a general factor, 2 = L^T of the
    // row-reversed factor of a reversed-order Cholesky (k_tile_body.h: flip): M[r][c] = 0 for r + c > n - 1
    if (i < nt && (dense == 1 || (dense == 0 && i <= cblk) || (dense == 2 && 16 * (i + cblk) <= n - 1))) {
      const int gr = 16 * cblk + lc;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int gc = 16 * i + lr + 4 * v;
        if (gr < n && gc < n) t[v] = Lmat[(size_t)gr * ld + gc];
      }
    }
    acc[r] = t;
#pragma unroll
    for (int u = 0; u < 3; ++u) ring[u][r] = double4_t{0.0, 0.0, 0.0, 0.0};
  });
  auto prefetch = [&](double4_t (&dst)[FW_ROWS], int k) {
    sfor<FW_ROWS>([&](auto rc) {
      constexpr int r = decltype(rc)::value;
      const int i = wave + 4 * r;
      if (i < ntl && i > k && k < ntl) {
        const double* tp = Ltp + (size_t)tile_index(i, k) * 256;
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[r][q] = tp[q * 64 + lane];
      }
    });
  };
  prefetch(ring[0], 0);
  prefetch(ring[1], 1);
  // the inverted diagonal blocks of this wave's rows, all of them up front (round 6): fetched at the start of the owner's step they
  // were one global round trip on the dependent chain of EVERY step
  double dqs[FW_ROWS][4];
  sfor<FW_ROWS>([&](auto rc) {
    constexpr int r = decltype(rc)::value;
    const int i = wave + 4 * r;
    const double* di = Dinv + (size_t)(i < ntl ? i : 0) * 256;
#pragma unroll
    for (int q = 0; q < 4; ++q) dqs[r][q] = di[lc * 16 + lr + 4 * q];
  });

  auto step = [&](int k, double4_t (&a_cur)[FW_ROWS], double4_t (&a_far)[FW_ROWS]) {
    double* vk = vt[k & 1];
    // diagonal solve by the owner of row k
    if ((k & 3) == wave) {
      double4_t vi = {0.0, 0.0, 0.0, 0.0};
      sfor<FW_ROWS>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        if (wave + 4 * r == k) {
          if (k < ntl) {
            // two independent accumulation chains of two instead of one of four (a dependent f64 MFMA starts ~200 cycles behind
            // its predecessor)
            double4_t vj = {0.0, 0.0, 0.0, 0.0};
            vi = __builtin_amdgcn_mfma_f64_16x16x4f64(dqs[r][0], acc[r][0], vi, 0, 0, 0);
            vj = __builtin_amdgcn_mfma_f64_16x16x4f64(dqs[r][1], acc[r][1], vj, 0, 0, 0);
            vi = __builtin_amdgcn_mfma_f64_16x16x4f64(dqs[r][2], acc[r][2], vi, 0, 0, 0);
            vj = __builtin_amdgcn_mfma_f64_16x16x4f64(dqs[r][3], acc[r][3], vj, 0, 0, 0);
            vi += vj;
          } else {
            vi = acc[r];  // behind the leading block Lt is the identity
          }
        }
      });
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int row = lr + 4 * v;
        vk[row * TS + lc] = vi[v];
        const int gr = 16 * k + row, gc = 16 * cblk + lc;
        if (gr < n && gc < n) V[(size_t)gr * ld + gc] = vi[v];
      }
    }
    prefetch(a_far, k + 2);
    // a barrier that orders LDS traffic only: __syncthreads() carries an s_waitcnt vmcnt(0), i.e. every step would wait for the
    // operand tiles just requested and for the stores of V; V_k went to vt[k & 1] (double buffered: the write of step k + 2 is two
    // barriers behind the reads of step k)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // trailing update of the rows i > k (increasing i: row k+1 first)
    double b[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) b[q] = vk[(lr + 4 * q) * TS + lc];
    sfor<FW_ROWS>([&](auto rc) {
      constexpr int r = decltype(rc)::value;
      const int i = wave + 4 * r;
      if (i < ntl && i > k && k < ntl) {
        double4_t t2 = {0.0, 0.0, 0.0, 0.0};  // (second chain, see the diagonal solve)
        acc[r] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a_cur[r][0], b[0], acc[r], 0, 0, 0);
        t2 = __builtin_amdgcn_mfma_f64_16x16x4f64(-a_cur[r][1], b[1], t2, 0, 0, 0);
        acc[r] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a_cur[r][2], b[2], acc[r], 0, 0, 0);
        t2 = __builtin_amdgcn_mfma_f64_16x16x4f64(-a_cur[r][3], b[3], t2, 0, 0, 0);
        acc[r] += t2;
      }
    });
  };
  for (int k = 0; k < nt; k += 3) {
    step(k, ring[0], ring[2]);
    if (k + 1 < nt) step(k + 1, ring[1], ring[0]);
    if (k + 2 < nt) step(k + 2, ring[2], ring[1]);
  }
}

// ------------------------------------------------------------------------------------------------
// GEMM, 4 waves per 16x16 output tile (split K, fixed-order LDS reduction), operands prefetched 4 k-steps deep.
// C = op(A) op(B) (+ I); optional symmetric mode computes only tiles with bi >= bj and mirrors them.
// ------------------------------------------------------------------------------------------------
template <bool TA, bool TB>
__global__ __launch_bounds__(256) void k_gemm4(int M, int N, int K, const double* __restrict__ A, int lda,
                                               const double* __restrict__ B, int ldb, double* __restrict__ C, int ldc,
                                               int add_identity, int symmetric, const int* __restrict__ cancel) {
  const int bi = blockIdx.y, bj = blockIdx.x;
  if (symmetric && bj > bi) return;
  if (cancel && *cancel) return;  // a factorization upstream failed: the destination (the resident covariance) stays as it was
  const int i0 = bi * 16, j0 = bj * 16;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int lr = lane >> 4, lc = lane & 15;
  const int ai = i0 + lc, bjj = j0 + lc;
  // wave w takes k-steps w, w+4, ... (one k-step = 4 consecutive k)
  const int nsteps = (K + 3) >> 2;
  double4_t acc = {0.0, 0.0, 0.0, 0.0};
  for (int st = wave; st < nsteps; st += 16) {
    double av[4], bv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = 4 * (st + 4 * u) + lr;
      av[u] = 0.0;
      bv[u] = 0.0;
      if (k < K && (st + 4 * u) < nsteps) {
        if (ai < M) av[u] = TA ? A[(size_t)k * lda + ai] : A[(size_t)ai * lda + k];
        if (bjj < N) bv[u] = TB ? B[(size_t)bjj * ldb + k] : B[(size_t)k * ldb + bjj];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
  }
  __shared__ double red[4][256];
#pragma unroll
  for (int v = 0; v < 4; ++v) red[wave][(lr + 4 * v) * 16 + lc] = acc[v];
  __syncthreads();
  const int row = tid >> 4, col = tid & 15;
  double sum = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
  const int gr = i0 + row, gc = j0 + col;
  if (gr < M && gc < N) {
    if (add_identity && gr == gc) sum += 1.0;
    C[(size_t)gr * ldc + gc] = sum;
    if (symmetric && bi != bj) C[(size_t)gc * ldc + gr] = sum;
  }
}

// dx = P b (one wave per row), negative-diagonal flag.  Optionally the LAST block to finish (ticket counter) publishes the
// result block [flags | dx] = `pub_words` 8-byte words into mapped pinned host memory, then the sequence word the host
// spins on, and clears the flags and the ticket for the next update - the update then ends without a separate publish launch.
// boost / boost_n / cancel: CholJob::boost - the amounts chol(P) added to the diagonal of the first boost_n columns come off the
// updated diagonal again (b is zero on those columns, so dx does not see them); nothing is touched when the update was
// cancelled (*cancel != 0: a failed factorization left the resident covariance alone)
__global__ __launch_bounds__(256) void k_dx_rows(double* P, int n, int ldp,
                                                  const double* __restrict__ b, double* __restrict__ dx,
                                                  int* __restrict__ negdiag, unsigned* __restrict__ ticket,
                                                  unsigned long long* __restrict__ res_block,
                                                  unsigned long long* __restrict__ host_block, int pub_words,
                                                  volatile unsigned* seq_host, unsigned seq, const double* __restrict__ boost,
                                                  int boost_n, const int* __restrict__ cancel) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row < n) {
    double* pr = P + (size_t)row * ldp;
    double s = 0.0;
    for (int c = lane; c < n; c += 64) s = fma(pr[c], b[c], s);
    s += xor_lane_f64<1>(s);  // (DPP + row swaps; wave_sum's six ds_bpermute round trips were the longest thing in this kernel)
    s += xor_lane_f64<2>(s);
    s += xor_lane_f64<4>(s);
    s += xor_lane_f64<8>(s);
    s = rows_sum_f64(s);
    if (lane == 0) {
      dx[row] = s;
      double d = pr[row];
      if (row < boost_n && !(cancel && *cancel)) {
        d -= boost[row];
        pr[row] = d;
      }
      if (d < 0.0) *negdiag = 1;
    }
  }
  if (!ticket) return;
  __shared__ unsigned last;
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) last = (atomicAdd(ticket, 1u) == gridDim.x - 1) ? 1u : 0u;
  __syncthreads();
  if (!last) return;
  __threadfence();
  for (int i = threadIdx.x; i < pub_words; i += 256) host_block[i] = __builtin_nontemporal_load(res_block + i);
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    *seq_host = seq;
    *ticket = 0u;
  }
  if (threadIdx.x < 2) res_block[threadIdx.x] = 0ull;  // the four flag words, cleared for the next update
}

}  // namespace ovp

extern "C" {

// returns hipErrorInvalidValue when n is too large for the register-resident path (caller falls back)
extern "C" { int ovp_dbg_tilechol_skip = 0; }
hipError_t ovp_launch_tilechol_unless(const double* A, double* L, double* Dinv, double* Lpack, int n, int ld, int* flag,
                                      int add_identity, const int* cond, hipStream_t stream);
hipError_t ovp_launch_tilechol(const double* A, double* L, double* Dinv, double* Lpack, int n, int ld, int* flag,
                               int add_identity, hipStream_t stream) {
  return ovp_launch_tilechol_unless(A, L, Dinv, Lpack, n, ld, flag, add_identity, nullptr, stream);
}
hipError_t ovp_launch_tilechol_unless(const double* A, double* L, double* Dinv, double* Lpack, int n, int ld, int* flag,
                                      int add_identity, const int* cond, hipStream_t stream) {
  const int nt = (n + 15) / 16;
  const int ntiles = nt * (nt + 1) / 2;
  const int slots = (ntiles + ovp::TC_TILE_WAVES - 1) / ovp::TC_TILE_WAVES;
  const size_t shmem = ((size_t)(2 + nt + ovp::TC_WAVES) * ovp::TSZ + 256) * sizeof(double) + 16;
  if (slots <= 15) {
    hipLaunchKernelGGL((ovp::k_tilechol<15>), dim3(1), dim3(ovp::TC_WAVES * 64), shmem, stream, A, L, Dinv, Lpack, n, ld,
                       flag, add_identity, ovp_dbg_tilechol_skip, cond);
  } else if (slots <= 18) {  // N <= 240: still without register spills
    hipLaunchKernelGGL((ovp::k_tilechol<18>), dim3(1), dim3(ovp::TC_WAVES * 64), shmem, stream, A, L, Dinv, Lpack, n, ld,
                       flag, add_identity, ovp_dbg_tilechol_skip, cond);
  } else if (slots <= 25) {
    static unsigned long long attr_mask = 0;  // per device (ovp_kernels.h)
    if (ovp_lds_attr_needed(&attr_mask)) {
      (void)hipFuncSetAttribute((const void*)ovp::k_tilechol<25>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
      (void)hipGetLastError();  // (a kernel with static LDS refuses the full 160 KB: harmless, a real shortage fails the launch itself)
      ovp_lds_attr_done(&attr_mask);
    }
    hipLaunchKernelGGL((ovp::k_tilechol<25>), dim3(1), dim3(ovp::TC_WAVES * 64), shmem, stream, A, L, Dinv, Lpack, n, ld,
                       flag, add_identity, ovp_dbg_tilechol_skip, cond);
  } else {
    return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t ovp_launch_fwdsub_lead(const double* Ltp, const double* Dinv, const double* Lmat, double* V, int n, int ld,
                                  int dense, int n_lead, hipStream_t stream) {
  const int nt = (n + 15) / 16;
  if (nt > 4 * ovp::FW_ROWS) return hipErrorInvalidValue;
  hipLaunchKernelGGL(ovp::k_fwdsub, dim3(nt), dim3(256), 0, stream, Ltp, Dinv, Lmat, V, n, ld, dense, n_lead);
  return hipGetLastError();
}
hipError_t ovp_launch_fwdsub(const double* Ltp, const double* Dinv, const double* Lmat, double* V, int n, int ld,
                             int dense, hipStream_t stream) {
  return ovp_launch_fwdsub_lead(Ltp, Dinv, Lmat, V, n, ld, dense, 0, stream);
}

hipError_t ovp_launch_gemm4c(int transA, int transB, int M, int N, int K, const double* A, int lda, const double* B, int ldb,
                             double* C, int ldc, int add_identity, int symmetric, const int* cancel, hipStream_t stream);
hipError_t ovp_launch_gemm4(int transA, int transB, int M, int N, int K, const double* A, int lda, const double* B,
                            int ldb, double* C, int ldc, int add_identity, int symmetric, hipStream_t stream) {
  return ovp_launch_gemm4c(transA, transB, M, N, K, A, lda, B, ldb, C, ldc, add_identity, symmetric, nullptr, stream);
}
// cancel: device flag; a non-zero value turns the product into a no-op (used for the write of the updated covariance)
hipError_t ovp_launch_gemm4c(int transA, int transB, int M, int N, int K, const double* A, int lda, const double* B, int ldb,
                             double* C, int ldc, int add_identity, int symmetric, const int* cancel, hipStream_t stream) {
  dim3 grid((N + 15) / 16, (M + 15) / 16), block(256);
  if (!transA && !transB)
    hipLaunchKernelGGL((ovp::k_gemm4<false, false>), grid, block, 0, stream, M, N, K, A, lda, B, ldb, C, ldc,
                       add_identity, symmetric, cancel);
  else if (transA && !transB)
    hipLaunchKernelGGL((ovp::k_gemm4<true, false>), grid, block, 0, stream, M, N, K, A, lda, B, ldb, C, ldc,
                       add_identity, symmetric, cancel);
  else if (!transA && transB)
    hipLaunchKernelGGL((ovp::k_gemm4<false, true>), grid, block, 0, stream, M, N, K, A, lda, B, ldb, C, ldc,
                       add_identity, symmetric, cancel);
  else
    hipLaunchKernelGGL((ovp::k_gemm4<true, true>), grid, block, 0, stream, M, N, K, A, lda, B, ldb, C, ldc,
                       add_identity, symmetric, cancel);
  return hipGetLastError();
}

hipError_t ovp_launch_dx_rows_boost(double* P, int n, int ldp, const double* b, double* dx, int* negdiag,
                                    unsigned* ticket, void* res_block, void* host_block, int pub_words, void* seq_host,
                                    unsigned seq, const double* boost, int boost_n, const int* cancel, hipStream_t stream) {
  hipLaunchKernelGGL(ovp::k_dx_rows, dim3((n + 3) / 4), dim3(256), 0, stream, P, n, ldp, b, dx, negdiag, ticket,
                     (unsigned long long*)res_block, (unsigned long long*)host_block, pub_words,
                     (volatile unsigned*)seq_host, seq, boost, boost_n, cancel);
  return hipGetLastError();
}
hipError_t ovp_launch_dx_rows(const double* P, int n, int ldp, const double* b, double* dx, int* negdiag,
                              unsigned* ticket, void* res_block, void* host_block, int pub_words, void* seq_host,
                              unsigned seq, hipStream_t stream) {
  return ovp_launch_dx_rows_boost(const_cast<double*>(P), n, ldp, b, dx, negdiag, ticket, res_block, host_block, pub_words, seq_host,
                                  seq, nullptr, 0, nullptr, stream);
}
}
