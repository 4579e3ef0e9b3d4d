// K3: EKF update from the information pair (A, b), covariance resident on the device, plus the covariance
// bookkeeping kernels (propagation, clone, marginalise, marginal gather).
//
// StateHelper::EKFUpdate (state/StateHelper.cpp:121-202) computes, for whitened noise R = I,
//     P+ = P - P H^T (H P H^T + I)^-1 H P ,   dx = P H^T (H P H^T + I)^-1 r .
// Both depend on H, r only through A = H^T H and b = H^T r:
//     P+ = (P^-1 + A)^-1 = L (I + L^T A L)^-1 L^T ,   dx = P+ b ,        with  P = L L^T .
// That form is SPD end to end (no P - K M^T cancellation, P+ symmetric PSD by construction):
//     L  = chol(P)            T = I + L^T (A L)          Lt = chol(T)
//     Y  = L Lt^-T            P+ = Y Y^T                 dx = P+ b
// The dense products run on v_mfma_f64_16x16x4_f64; the two factorizations are latency-bound single-workgroup
// kernels (N ~ 200-400), chol(P) is independent of the measurements and is overlapped with K1/K2 by the host.
#include "ovp_dev.h"
#include "ovp_kernels.h"

namespace ovp {

typedef double double4_t __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// Blocked right-looking Cholesky, one workgroup of 1024 threads, panel in LDS.
// W (n x n, ld) holds the symmetric input on entry (only the lower triangle is read) and L on exit
// (strict upper triangle zeroed).
// ------------------------------------------------------------------------------------------------
static constexpr int CH_NB = 32;
static constexpr int CH_T = 1024;

__global__ __launch_bounds__(CH_T) void k_chol(const double* __restrict__ A, double* __restrict__ W, int n, int ld,
                                               int* __restrict__ flag, int add_identity) {
  extern __shared__ __attribute__((aligned(16))) double panel[];  // [n][CH_NB+1]
  const int t = threadIdx.x;
  const int PL = CH_NB + 1;
  // copy lower triangle of A (+I) into W, zero the strict upper triangle
  for (int idx = t; idx < n * n; idx += CH_T) {
    const int i = idx / n, j = idx - i * n;
    double v = 0.0;
    if (j <= i) {
      v = A[(size_t)i * ld + j];
      if (add_identity && i == j) v += 1.0;
    }
    W[(size_t)i * ld + j] = v;
  }
  __syncthreads();
  for (int kb = 0; kb < n; kb += CH_NB) {
    const int nb = min(CH_NB, n - kb);
    const int mrem = n - kb;
    // load panel rows kb..n-1, columns kb..kb+nb-1
    for (int idx = t; idx < mrem * nb; idx += CH_T) {
      const int i = idx / nb, j = idx - i * nb;
      panel[i * PL + j] = W[(size_t)(kb + i) * ld + kb + j];
    }
    __syncthreads();
    // factor the panel column by column (left-looking inside the panel)
    for (int j = 0; j < nb; ++j) {
      const double d = panel[j * PL + j];
      if (t == 0 && !(d > 0.0)) *flag = 1;
      const double inv = 1.0 / sqrt(d);
      __syncthreads();
      // scale column j (rows j..mrem-1); row j gets sqrt(d)
      for (int i = j + t; i < mrem; i += CH_T) panel[i * PL + j] *= inv;
      __syncthreads();
      // update the remaining panel columns c in (j, nb): panel[i][c] -= panel[i][j] * panel[c][j], i >= c
      const int ncol = nb - j - 1;
      if (ncol > 0) {
        const int nrow = mrem - j - 1;
        for (int idx = t; idx < nrow * ncol; idx += CH_T) {
          const int i = j + 1 + idx / ncol, c = j + 1 + (idx % ncol);
          if (i >= c) panel[i * PL + c] -= panel[i * PL + j] * panel[c * PL + j];
        }
      }
      __syncthreads();
    }
    // write the factored panel back
    for (int idx = t; idx < mrem * nb; idx += CH_T) {
      const int i = idx / nb, j = idx - i * nb;
      W[(size_t)(kb + i) * ld + kb + j] = (j <= i) ? panel[i * PL + j] : 0.0;
    }
    // trailing update: W[i][j] -= sum_c panel[i][c] panel[j][c],  i >= j >= kb+nb
    const int tr0 = nb;  // first trailing row inside the panel
    const int ntr = mrem - nb;
    if (ntr > 0) {
      // 2D thread tiling: each thread handles a 2x2 micro-tile walk over the lower triangle
      for (int idx = t; idx < ntr * ntr; idx += CH_T) {
        const int i = idx / ntr, j = idx - i * ntr;
        if (j <= i) {
          double s = 0.0;
          const double* pi = panel + (tr0 + i) * PL;
          const double* pj = panel + (tr0 + j) * PL;
#pragma unroll 8
          for (int c = 0; c < nb; ++c) s = fma(pi[c], pj[c], s);
          W[(size_t)(kb + nb + i) * ld + kb + nb + j] -= s;
        }
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// Small dense GEMM on v_mfma_f64_16x16x4_f64: one wave per 16x16 tile of C = op(A) op(B) (+ I).
// ------------------------------------------------------------------------------------------------
template <bool TA, bool TB>
__global__ __launch_bounds__(64) void k_gemm(int M, int N, int K, const double* __restrict__ A, int lda,
                                             const double* __restrict__ B, int ldb, double* __restrict__ C, int ldc,
                                             int add_identity) {
  const int i0 = blockIdx.y * 16, j0 = blockIdx.x * 16;
  const int lane = threadIdx.x;
  const int kk = lane >> 4, ij = lane & 15;
  const int ai = i0 + ij, bj = j0 + ij;
  double4_t acc = {0.0, 0.0, 0.0, 0.0};
  for (int k0 = 0; k0 < K; k0 += 4) {
    const int k = k0 + kk;
    double av = 0.0, bv = 0.0;
    if (k < K) {
      if (ai < M) av = TA ? A[(size_t)k * lda + ai] : A[(size_t)ai * lda + k];
      if (bj < N) bv = TB ? B[(size_t)bj * ldb + k] : B[(size_t)k * ldb + bj];
    }
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
  }
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int row = i0 + (lane >> 4) + 4 * v, col = j0 + (lane & 15);
    if (row < M && col < N) C[(size_t)row * ldc + col] = acc[v] + ((add_identity && row == col) ? 1.0 : 0.0);
  }
}

// ------------------------------------------------------------------------------------------------
// Y Lt^T = L  (Lt lower triangular): every row y of Y solves Lt y^T = l^T by forward substitution.
// block = 256 threads = 16 rows x 16 column lanes; the 16 lanes of a row live in one wave.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_trsm_right_lt(const double* __restrict__ L, const double* __restrict__ Lt,
                                                        double* __restrict__ Y, int n, int ld) {
  extern __shared__ __attribute__((aligned(16))) double ybuf[];  // [16][n]
  const int rl = threadIdx.x >> 4, cl = threadIdx.x & 15;
  const int row = blockIdx.x * 16 + rl;
  double* y = ybuf + (size_t)rl * n;
  const bool rv = row < n;
  for (int j = 0; j < n; ++j) {
    const double* ltj = Lt + (size_t)j * ld;
    double s = 0.0;
    for (int p = cl; p < j; p += 16) s = fma(ltj[p], y[p], s);
    // reduce over the 16 lanes of this row
    s += shfl_xor_f64(s, 8);
    s += shfl_xor_f64(s, 4);
    s += shfl_xor_f64(s, 2);
    s += shfl_xor_f64(s, 1);
    if (cl == 0) {
      const double lij = rv ? L[(size_t)row * ld + j] : 0.0;
      y[j] = (lij - s) / ltj[j];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
  if (rv)
    for (int j = cl; j < n; j += 16) Y[(size_t)row * ld + j] = y[j];
}

// dx = P b ; negdiag flag
__global__ void k_dx_negdiag(const double* __restrict__ P, int n, int ldp, const double* __restrict__ b,
                             double* __restrict__ dx, int* __restrict__ negdiag) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const double* pr = P + (size_t)r * ldp;
  double s = 0.0;
  for (int c = 0; c < n; ++c) s = fma(pr[c], b[c], s);
  dx[r] = s;
  if (pr[r] < 0.0) *negdiag = 1;
}

// ------------------------------------------------------------------------------------------------
// covariance bookkeeping (keeps State::_Cov resident)
// ------------------------------------------------------------------------------------------------
// StateHelper::get_marginal_covariance: out[i*m + k] = P[cols[i]][cols[k]]
__global__ void k_gather_marginal(const double* __restrict__ P, int ldp, const int* __restrict__ cols, int m,
                                  double* __restrict__ out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
  if (k < m) out[(size_t)i * m + k] = P[(size_t)cols[i] * ldp + cols[k]];
}

// Sub-state update (N above the tile factorization's limit): out[i*ldo + k] = P[ids[i]][ids[k]] ; G[r*ldg + k] = P[r][ids[k]] ;
// C = A - B ; P -= D.
__global__ void k_gather_block(const double* __restrict__ P, int ldp, const int* __restrict__ ids, int m,
                               double* __restrict__ out, int ldo) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
  if (k < m) out[(size_t)i * ldo + k] = P[(size_t)ids[i] * ldp + ids[k]];
}
// the same with a relative boost of the diagonal from position `from` on: out[i][i] = (1 + rel) P[ids[i]][ids[i]], the added amounts
// left in boost[state column] (zero for the columns in front of `from`).  The plane loop's order ends with the columns no plane
// of the call involves (IMU, dt, ...): every A_k is zero there, so the loop returns exactly P+ + diag(boost) on them
// ((P + D)+ = P+ + D when H D = 0) and k_unpermute_pair takes the amounts off again - and an exact stochastic clone (IMU pose ==
// newest clone: the conditional covariance of the IMU pose given the clones is exactly zero) factors without a second attempt.
__global__ void k_gather_block_boost(const double* __restrict__ P, int ldp, const int* __restrict__ ids, int m,
                                     double* __restrict__ out, int ldo, int from, double rel, double* __restrict__ boost) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
  if (k >= m) return;
  double v = P[(size_t)ids[i] * ldp + ids[k]];
  if (i == k) {
    const double add = i >= from ? v * rel : 0.0;
    boost[ids[i]] = add;
    v += add;
  }
  out[(size_t)i * ldo + k] = v;
}
// Positive SEMI-definite covariance in front of a pivot-dropping factorization: C = D^-1 P D^-1 with D = sqrt(diag P) (unit diagonal,
// so ONE absolute pivot floor separates the directions P does not determine - pivots of rounding size - from the regular ones,
// whatever the units of the variables); a variable with zero variance gets a zero row / column.  dvec <- D.
__global__ void k_unit_diag(const double* __restrict__ P, int n, int ld, double* __restrict__ C, double* __restrict__ dvec) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
  if (k >= n) return;
  const double pi = P[(size_t)i * ld + i], pk = P[(size_t)k * ld + k];
  const double si = pi > 0.0 ? 1.0 / sqrt(pi) : 0.0, sk = pk > 0.0 ? 1.0 / sqrt(pk) : 0.0;
  C[(size_t)i * ld + k] = (i == k) ? (pi > 0.0 ? 1.0 : 0.0) : P[(size_t)i * ld + k] * si * sk;
  if (i == 0) dvec[k] = pk > 0.0 ? sqrt(pk) : 0.0;
}
// L <- D L (rows scaled): the factor of P from the factor of its unit-diagonal form
__global__ void k_scale_rows(double* __restrict__ L, int n, int ld, const double* __restrict__ dvec) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
  if (k < n) L[(size_t)i * ld + k] *= dvec[i];
}
// out[r][k] = V[k][ids ? ids[r] : r]: the (dense) factor M = V^T of P = V^T V, rows back in the state's own order when V was
// formed in a permuted one (plane loop) - any M with M M^T = P serves the next update's  P+ = M (I + M^T A M)^-1 M^T
__global__ void k_factor_from_V(const double* __restrict__ V, int ld, const int* __restrict__ ids, int n, double* __restrict__ out,
                                int ldo) {
  __shared__ double tile[16][17];
  const int r0 = blockIdx.y * 16, k0 = blockIdx.x * 16;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  // read V[k0 + ty][col(r0 + tx)] (rows of V contiguous in tx when the order is the identity), write out[r0 + ty][k0 + tx]
  const int r = r0 + tx, k = k0 + ty;
  double v = 0.0;
  if (r < n && k < n) v = V[(size_t)k * ld + (ids ? ids[r] : r)];
  tile[ty][tx] = v;
  __syncthreads();
  const int ro = r0 + ty, ko = k0 + tx;
  if (ro < n && ko < n) out[(size_t)ro * ldo + ko] = tile[tx][ty];
}
// Both of the above behind a plane loop that ran in its own column order, in one launch: Pout[r][k] = Pperm[ids[r]][ids[k]]
// (unless *cancel: a failed factorization leaves the resident covariance alone) and Lout[r][k] = V[k][ids[r]].
__global__ void k_unpermute_pair(const double* __restrict__ Pperm, const double* __restrict__ V, int ld, const int* __restrict__ ids,
                                 int n, double* __restrict__ Pout, double* __restrict__ Lout, int ldo, const int* __restrict__ cancel,
                                 const double* __restrict__ boost) {
  __shared__ double tile[16][17];
  const int r0 = blockIdx.y * 16, k0 = blockIdx.x * 16;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int r = r0 + tx, k = k0 + ty;
  double v = 0.0;
  if (r < n && k < n) v = V[(size_t)k * ld + ids[r]];
  tile[ty][tx] = v;
  __syncthreads();
  const int ro = r0 + ty, ko = k0 + tx;
  if (ro < n && ko < n) {
    if (Lout) Lout[(size_t)ro * ldo + ko] = tile[tx][ty];
    if (*cancel == 0)
      Pout[(size_t)ro * ldo + ko] = Pperm[(size_t)ids[ro] * ld + ids[ko]] - ((boost && ro == ko) ? boost[ro] : 0.0);  // (k_gather_block_boost)
  }
}
// ... unless *cancel != 0 (a failed factorization upstream: the destination keeps what it holds, cf. ovp_launch_gemm4c)
__global__ void k_gather_block_unless(const double* __restrict__ P, int ldp, const int* __restrict__ ids, int m,
                                      double* __restrict__ out, int ldo, const int* __restrict__ cancel) {
  if (*cancel != 0) return;
  const int k = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
  if (k < m) out[(size_t)i * ldo + k] = P[(size_t)ids[i] * ldp + ids[k]];
}
__global__ void k_gather_cols(const double* __restrict__ P, int ldp, const int* __restrict__ ids, int n, int m,
                              double* __restrict__ G, int ldg) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
  if (k < m && r < n) G[(size_t)r * ldg + k] = P[(size_t)r * ldp + ids[k]];
}
__global__ void k_mat_sub(const double* __restrict__ A, const double* __restrict__ B, double* __restrict__ C, int rows, int cols,
                          int ld) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
  if (k < cols && i < rows) C[(size_t)i * ld + k] = A[(size_t)i * ld + k] - B[(size_t)i * ld + k];
}
// P -= D on the lower triangle, mirrored (the reference mirrors its upper triangle the same way, StateHelper.cpp:171-172)
__global__ void k_sub_sym(double* __restrict__ P, const double* __restrict__ D, int n, int ld, const int* __restrict__ cancel) {
  if (cancel && *cancel != 0) return;  // a factorization upstream failed: the covariance keeps what it holds
  const int k = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
  if (k <= i && i < n) {
    const double v = P[(size_t)i * ld + k] - D[(size_t)i * ld + k];
    P[(size_t)i * ld + k] = v;
    P[(size_t)k * ld + i] = v;
  }
}

// StateHelper::clone: P[new..new+sz) rows/cols = copies of [src..src+sz)
__global__ void k_cov_clone(double* __restrict__ P, int ldp, int n_old, int src, int sz, double jitter) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int n_new = n_old + sz;
  if (idx >= n_new * sz) return;
  const int r = idx / sz, k = idx - r * sz;  // r over all (new) rows, k over the new columns
  if (r < n_old) {
    // Cov.block(0,new,old,sz) = Cov.block(0,old_loc,...) ; Cov.block(new,0,sz,old) = Cov.block(old_loc,0,...)
    P[(size_t)r * ldp + n_old + k] = P[(size_t)r * ldp + src + k];
    P[(size_t)(n_old + k) * ldp + r] = P[(size_t)(src + k) * ldp + r];
  } else {
    double v = P[(size_t)(src + (r - n_old)) * ldp + src + k];
    // The clone is an exact copy (state/StateHelper.cpp:346-396), so P is only positive SEMI-definite from here until a
    // propagation puts process noise on the source; the update entry points then leave chol(P) for their pivot-dropping / S-form
    // paths.  jitter > 0 (ovp_cov_clone_jitter, off by default) stores the diagonal of the new block (1 + jitter) x the copied
    // value instead, which keeps such a prior on the fast path at the price of that departure from the reference.
    if (r - n_old == k) v *= (1.0 + jitter);
    P[(size_t)r * ldp + n_old + k] = v;
  }
}

// StateHelper::marginalize: compact rows/cols [id, id+sz) out of src into dst
__global__ void k_cov_marginalize(const double* __restrict__ src, double* __restrict__ dst, int ld, int n_old, int id,
                                  int sz) {
  const int n_new = n_old - sz;
  const int c = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
  if (c >= n_new) return;
  const int rs = r < id ? r : r + sz, cs = c < id ? c : c + sz;
  dst[(size_t)r * ld + c] = src[(size_t)rs * ld + cs];
}

// StateHelper::EKFPropagation, step 1: CPT[r][j] = sum_k P[r][oldcol[k]] Phi[j][k]   (n x phi)
__global__ void k_prop_cpt(const double* __restrict__ P, int ldp, int n, const int* __restrict__ oldcol, int nold,
                           const double* __restrict__ Phi, int phi, double* __restrict__ CPT) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
  if (j >= phi) return;
  double s = 0.0;
  for (int k = 0; k < nold; ++k) s = fma(P[(size_t)r * ldp + oldcol[k]], Phi[(size_t)k * phi + j], s);  // Phi col-major
  CPT[(size_t)r * phi + j] = s;
}
// step 2: PCP[i][j] = Qsym[i][j] + sum_k Phi[i][k] CPT[oldcol[k]][j]
__global__ void k_prop_pcp(const double* __restrict__ CPT, const int* __restrict__ oldcol, int nold,
                           const double* __restrict__ Phi, const double* __restrict__ Q, int phi,
                           double* __restrict__ PCP) {
  const int j = threadIdx.x, i = blockIdx.x;
  if (j >= phi) return;
  double s = (i <= j) ? Q[(size_t)j * phi + i] : Q[(size_t)i * phi + j];  // Q col-major, upper triangle read
  for (int k = 0; k < nold; ++k) s = fma(Phi[(size_t)k * phi + i], CPT[(size_t)oldcol[k] * phi + j], s);
  PCP[(size_t)i * phi + j] = s;
}
// step 3: write the row strip, column strip and diagonal block; negative-diagonal check
__global__ void k_prop_write(double* __restrict__ P, int ldp, int n, int start, int phi, const double* __restrict__ CPT,
                             const double* __restrict__ PCP, int* __restrict__ negdiag) {
  const int j = threadIdx.x, r = blockIdx.x;
  if (j >= phi) return;
  const bool inblk = (r >= start && r < start + phi);
  const double v = inblk ? PCP[(size_t)(r - start) * phi + j] : CPT[(size_t)r * phi + j];
  P[(size_t)r * ldp + start + j] = v;
  if (!inblk) P[(size_t)(start + j) * ldp + r] = v;
  if (inblk && (r - start) == j && v < 0.0) *negdiag = 1;
}
// StateHelper::augment_clone time-offset Jacobian, step 1 (columns) and step 2 (rows, reads the updated row dt)
__global__ void k_augment_dt_cols(double* __restrict__ P, int ldp, int n, int pose, int dt, double d0, double d1, double d2,
                                  double d3, double d4, double d5) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const double v = P[(size_t)r * ldp + dt];
  const double d[6] = {d0, d1, d2, d3, d4, d5};
  for (int k = 0; k < 6; ++k) P[(size_t)r * ldp + pose + k] += v * d[k];
}
__global__ void k_augment_dt_rows(double* __restrict__ P, int ldp, int n, int pose, int dt, double d0, double d1, double d2,
                                  double d3, double d4, double d5) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  const double v = P[(size_t)dt * ldp + c];
  const double d[6] = {d0, d1, d2, d3, d4, d5};
  for (int k = 0; k < 6; ++k) P[(size_t)(pose + k) * ldp + c] += d[k] * v;
}
// StateHelper::initialize_invertible (state/StateHelper.cpp:520-573), k <= 6 new columns.
// step 1: M_a[r][j] = sum_a P[r][cols[a]] * H_R[j][a]          (H_R row-major [k][ncols] on the device)
__global__ void k_init_ma(const double* __restrict__ P, int ldp, int n, const int* __restrict__ cols, int ncols,
                          const double* __restrict__ HR, int k, double* __restrict__ Ma) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  double acc[6] = {0, 0, 0, 0, 0, 0};
  for (int a = 0; a < ncols; ++a) {
    const double pv = P[(size_t)r * ldp + cols[a]];
#pragma unroll
    for (int j = 0; j < 6; ++j)
      if (j < k) acc[j] = fma(pv, HR[(size_t)j * ncols + a], acc[j]);
  }
#pragma unroll
  for (int j = 0; j < 6; ++j)
    if (j < k) Ma[(size_t)r * 6 + j] = acc[j];
}
// step 2 (one workgroup): M = H_R M_a[cols] + R (upper triangle mirrored), P_LL = Hinv M Hinv^T,
// P[0:n, n:n+k] = -M_a Hinv^T and its transpose.   Hinv, Rk are k x k row-major.
__global__ __launch_bounds__(256) void k_init_write(double* __restrict__ P, int ldp, int n, const int* __restrict__ cols,
                                                     int ncols, const double* __restrict__ HR, int k,
                                                     const double* __restrict__ Ma, const double* __restrict__ Hinv,
                                                     const double* __restrict__ Rk) {
  __shared__ double M[36], PLL[36], Hi[36];
  const int t = threadIdx.x;
  if (t < k * k) {
    const int i = t / k, j = t - i * k;
    Hi[t] = Hinv[t];
    const int ii = i <= j ? i : j, jj = i <= j ? j : i;  // selfadjointView<Upper>
    double s = Rk[ii * k + jj];
    for (int a = 0; a < ncols; ++a) s = fma(HR[(size_t)ii * ncols + a], Ma[(size_t)cols[a] * 6 + jj], s);
    M[t] = s;
  }
  __syncthreads();
  if (t < k * k) {
    const int i = t / k, j = t - i * k;
    double s = 0.0;
    for (int a = 0; a < k; ++a)
      for (int b = 0; b < k; ++b) s = fma(Hi[i * k + a] * M[a * k + b], Hi[j * k + b], s);
    PLL[t] = s;
  }
  __syncthreads();
  for (int idx = t; idx < n * k; idx += 256) {
    const int r = idx / k, j = idx - r * k;
    double s = 0.0;
    for (int a = 0; a < k; ++a) s = fma(Ma[(size_t)r * 6 + a], Hi[j * k + a], s);
    P[(size_t)r * ldp + n + j] = -s;
    P[(size_t)(n + j) * ldp + r] = -s;
  }
  if (t < k * k) {
    const int i = t / k, j = t - i * k;
    P[(size_t)(n + i) * ldp + n + j] = PLL[t];
  }
}

__global__ void k_check_negdiag(const double* __restrict__ P, int ldp, int n, int* __restrict__ negdiag) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n && P[(size_t)r * ldp + r] < 0.0) *negdiag = 1;
}

// The same check as the LAST kernel of a propagation, in one block: the verdict goes straight into mapped pinned memory, then a
// sequence word the host spins on (host[0] = seq, host[1] = verdict), and the device word is cleared for the next user - instead of
// a D2H copy command and a stream synchronisation behind a 2 us kernel (propagation step 42 -> 29 us, bench: propagation_cov_step_us).
__global__ __launch_bounds__(1024) void k_check_negdiag_publish(const double* __restrict__ P, int ldp, int n, int* __restrict__ negdiag,
                                                               volatile unsigned* host, unsigned seq) {
  __shared__ int any;
  if (threadIdx.x == 0) any = 0;
  __syncthreads();
  for (int r = threadIdx.x; r < n; r += 1024)
    if (P[(size_t)r * ldp + r] < 0.0) any = 1;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int v = (any | *negdiag) ? 1 : 0;
    *negdiag = 0;
    host[1] = (unsigned)v;
    __threadfence_system();
    host[0] = seq;
  }
}

}  // namespace ovp

extern "C" {

hipError_t ovp_launch_chol(const double* A, double* L, int n, int ld, int* flag, int add_identity,
                           hipStream_t stream) {
  const size_t shmem = (size_t)n * (ovp::CH_NB + 1) * sizeof(double);
  if (shmem > 160 * 1024) return hipErrorInvalidValue;
  static unsigned long long attr_mask = 0;  // per device (ovp_kernels.h)
  if (ovp_lds_attr_needed(&attr_mask)) {
    (void)hipFuncSetAttribute((const void*)ovp::k_chol, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipGetLastError();  // (a kernel with static LDS refuses the full 160 KB: harmless, a real shortage fails the launch itself)
    ovp_lds_attr_done(&attr_mask);
  }
  hipLaunchKernelGGL(ovp::k_chol, dim3(1), dim3(ovp::CH_T), shmem, stream, A, L, n, ld, flag, add_identity);
  return hipGetLastError();
}

hipError_t ovp_launch_gemm(int transA, int transB, int M, int N, int K, const double* A, int lda, const double* B,
                           int ldb, double* C, int ldc, int add_identity, hipStream_t stream) {
  dim3 grid((N + 15) / 16, (M + 15) / 16), block(64);
  if (!transA && !transB)
    hipLaunchKernelGGL((ovp::k_gemm<false, false>), grid, block, 0, stream, M, N, K, A, lda, B, ldb, C, ldc, add_identity);
  else if (transA && !transB)
    hipLaunchKernelGGL((ovp::k_gemm<true, false>), grid, block, 0, stream, M, N, K, A, lda, B, ldb, C, ldc, add_identity);
  else if (!transA && transB)
    hipLaunchKernelGGL((ovp::k_gemm<false, true>), grid, block, 0, stream, M, N, K, A, lda, B, ldb, C, ldc, add_identity);
  else
    hipLaunchKernelGGL((ovp::k_gemm<true, true>), grid, block, 0, stream, M, N, K, A, lda, B, ldb, C, ldc, add_identity);
  return hipGetLastError();
}

hipError_t ovp_launch_trsm_right_lt(const double* L, const double* Lt, double* Y, int n, int ld, hipStream_t stream) {
  const size_t shmem = (size_t)16 * n * sizeof(double);
  hipLaunchKernelGGL(ovp::k_trsm_right_lt, dim3((n + 15) / 16), dim3(256), shmem, stream, L, Lt, Y, n, ld);
  return hipGetLastError();
}

hipError_t ovp_launch_cov_finish(const double* Y, int n, int ld, const double* b, double* P, int ldp, double* dx,
                                 int* negdiag, hipStream_t stream) {
  hipError_t e = ovp_launch_gemm(0, 1, n, n, n, Y, ld, Y, ld, P, ldp, 0, stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(ovp::k_dx_negdiag, dim3((n + 63) / 64), dim3(64), 0, stream, P, n, ldp, b, dx, negdiag);
  return hipGetLastError();
}

hipError_t ovp_launch_gather_marginal(const double* P, int ldp, const int* cols, int m, double* out,
                                      hipStream_t stream) {
  hipLaunchKernelGGL(ovp::k_gather_marginal, dim3((m + 127) / 128, m), dim3(128), 0, stream, P, ldp, cols, m, out);
  return hipGetLastError();
}

hipError_t ovp_launch_gather_block(const double* P, int ldp, const int* ids, int m, double* out, int ldo, hipStream_t stream) {
  hipLaunchKernelGGL(ovp::k_gather_block, dim3((m + 127) / 128, m), dim3(128), 0, stream, P, ldp, ids, m, out, ldo);
  return hipGetLastError();
}
hipError_t ovp_launch_gather_block_boost(const double* P, int ldp, const int* ids, int m, double* out, int ldo, int from, double rel,
                                         double* boost, hipStream_t stream) {
  hipLaunchKernelGGL(ovp::k_gather_block_boost, dim3((m + 127) / 128, m), dim3(128), 0, stream, P, ldp, ids, m, out, ldo, from, rel, boost);
  return hipGetLastError();
}
hipError_t ovp_launch_factor_from_V(const double* V, int ld, const int* ids, int n, double* out, int ldo, hipStream_t stream) {
  const int nt = (n + 15) / 16;
  hipLaunchKernelGGL(ovp::k_factor_from_V, dim3(nt, nt), dim3(256), 0, stream, V, ld, ids, n, out, ldo);
  return hipGetLastError();
}
hipError_t ovp_launch_unpermute_pair(const double* Pperm, const double* V, int ld, const int* ids, int n, double* Pout, double* Lout,
                                     int ldo, const int* cancel, const double* boost, hipStream_t stream) {
  const int nt = (n + 15) / 16;
  hipLaunchKernelGGL(ovp::k_unpermute_pair, dim3(nt, nt), dim3(256), 0, stream, Pperm, V, ld, ids, n, Pout, Lout, ldo, cancel, boost);
  return hipGetLastError();
}
hipError_t ovp_launch_unit_diag(const double* P, int n, int ld, double* C, double* dvec, hipStream_t stream) {
  hipLaunchKernelGGL(ovp::k_unit_diag, dim3((n + 127) / 128, n), dim3(128), 0, stream, P, n, ld, C, dvec);
  return hipGetLastError();
}
hipError_t ovp_launch_scale_rows(double* L, int n, int ld, const double* dvec, hipStream_t stream) {
  hipLaunchKernelGGL(ovp::k_scale_rows, dim3((n + 127) / 128, n), dim3(128), 0, stream, L, n, ld, dvec);
  return hipGetLastError();
}
hipError_t ovp_launch_gather_block_unless(const double* P, int ldp, const int* ids, int m, double* out, int ldo, const int* cancel,
                                          hipStream_t stream) {
  hipLaunchKernelGGL(ovp::k_gather_block_unless, dim3((m + 127) / 128, m), dim3(128), 0, stream, P, ldp, ids, m, out, ldo, cancel);
  return hipGetLastError();
}
hipError_t ovp_launch_gather_cols(const double* P, int ldp, const int* ids, int n, int m, double* G, int ldg, hipStream_t stream) {
  hipLaunchKernelGGL(ovp::k_gather_cols, dim3((m + 127) / 128, n), dim3(128), 0, stream, P, ldp, ids, n, m, G, ldg);
  return hipGetLastError();
}
hipError_t ovp_launch_sub_sym_unless(double* P, const double* D, int n, int ld, const int* cancel, hipStream_t stream) {
  hipLaunchKernelGGL(ovp::k_sub_sym, dim3((n + 127) / 128, n), dim3(128), 0, stream, P, D, n, ld, cancel);
  return hipGetLastError();
}
hipError_t ovp_launch_sub_sym(double* P, const double* D, int n, int ld, hipStream_t stream) {
  return ovp_launch_sub_sym_unless(P, D, n, ld, nullptr, stream);
}
hipError_t ovp_launch_mat_sub(const double* A, const double* B, double* C, int rows, int cols, int ld, hipStream_t stream) {
  hipLaunchKernelGGL(ovp::k_mat_sub, dim3((cols + 127) / 128, rows), dim3(128), 0, stream, A, B, C, rows, cols, ld);
  return hipGetLastError();
}

hipError_t ovp_launch_cov_clone(double* P, int ldp, int n_old, int src, int sz, double jitter, hipStream_t stream) {
  const int total = (n_old + sz) * sz;
  hipLaunchKernelGGL(ovp::k_cov_clone, dim3((total + 255) / 256), dim3(256), 0, stream, P, ldp, n_old, src, sz, jitter);
  return hipGetLastError();
}

hipError_t ovp_launch_cov_marginalize(const double* src, double* dst, int ld, int n_old, int id, int sz,
                                      hipStream_t stream) {
  const int n_new = n_old - sz;
  if (n_new <= 0) return hipSuccess;
  hipLaunchKernelGGL(ovp::k_cov_marginalize, dim3((n_new + 127) / 128, n_new), dim3(128), 0, stream, src, dst, ld,
                     n_old, id, sz);
  return hipGetLastError();
}

hipError_t ovp_launch_init_invertible(double* P, int ldp, int n, const int* cols, int ncols, const double* HR, int k,
                                      double* Ma, const double* Hinv, const double* Rk, hipStream_t stream) {
  hipLaunchKernelGGL(ovp::k_init_ma, dim3((n + 127) / 128), dim3(128), 0, stream, P, ldp, n, cols, ncols, HR, k, Ma);
  hipLaunchKernelGGL(ovp::k_init_write, dim3(1), dim3(256), 0, stream, P, ldp, n, cols, ncols, HR, k, Ma, Hinv, Rk);
  return hipGetLastError();
}

hipError_t ovp_launch_augment_dt(double* P, int ldp, int n, int pose, int dt, const double* d, hipStream_t stream) {
  hipLaunchKernelGGL(ovp::k_augment_dt_cols, dim3((n + 127) / 128), dim3(128), 0, stream, P, ldp, n, pose, dt, d[0], d[1], d[2],
                     d[3], d[4], d[5]);
  hipLaunchKernelGGL(ovp::k_augment_dt_rows, dim3((n + 127) / 128), dim3(128), 0, stream, P, ldp, n, pose, dt, d[0], d[1], d[2],
                     d[3], d[4], d[5]);
  return hipGetLastError();
}

hipError_t ovp_launch_propagate(double* P, int ldp, int n, int start, int phi, const int* oldcol, int nold,
                                const double* Phi, const double* Q, double* CPT, double* PCP, int* negdiag,
                                hipStream_t stream) {
  hipLaunchKernelGGL(ovp::k_prop_cpt, dim3((phi + 63) / 64, n), dim3(64), 0, stream, P, ldp, n, oldcol, nold, Phi, phi,
                     CPT);
  hipLaunchKernelGGL(ovp::k_prop_pcp, dim3(phi), dim3(((phi + 63) / 64) * 64), 0, stream, CPT, oldcol, nold, Phi, Q, phi,
                     PCP);
  hipLaunchKernelGGL(ovp::k_prop_write, dim3(n), dim3(((phi + 63) / 64) * 64), 0, stream, P, ldp, n, start, phi, CPT,
                     PCP, negdiag);
  hipLaunchKernelGGL(ovp::k_check_negdiag, dim3((n + 255) / 256), dim3(256), 0, stream, P, ldp, n, negdiag);
  return hipGetLastError();
}
// ... with the verdict published to mapped pinned memory by the last kernel (k_check_negdiag_publish)
hipError_t ovp_launch_propagate_publish(double* P, int ldp, int n, int start, int phi, const int* oldcol, int nold, const double* Phi,
                                        const double* Q, double* CPT, double* PCP, int* negdiag, unsigned* host_dev, unsigned seq,
                                        hipStream_t stream) {
  hipLaunchKernelGGL(ovp::k_prop_cpt, dim3((phi + 63) / 64, n), dim3(64), 0, stream, P, ldp, n, oldcol, nold, Phi, phi,
                     CPT);
  hipLaunchKernelGGL(ovp::k_prop_pcp, dim3(phi), dim3(((phi + 63) / 64) * 64), 0, stream, CPT, oldcol, nold, Phi, Q, phi,
                     PCP);
  hipLaunchKernelGGL(ovp::k_prop_write, dim3(n), dim3(((phi + 63) / 64) * 64), 0, stream, P, ldp, n, start, phi, CPT,
                     PCP, negdiag);
  hipLaunchKernelGGL(ovp::k_check_negdiag_publish, dim3(1), dim3(1024), 0, stream, P, ldp, n, negdiag, (volatile unsigned*)host_dev, seq);
  return hipGetLastError();
}
}
