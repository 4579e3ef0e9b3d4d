// K2: reduce the per-feature outputs of K1 to the information pair  A = sum_f Hp_f^T Hp_f,  b = sum_f Hp_f^T r_f.
//
// This is the MI355X-native replacement of the reference's stacking + measurement compression
// (update/UpdaterMSCKF.cpp:767-805, update/UpdaterHelper.cpp:548-579): instead of Givens-rotating a
// (sum(2m-3) x c) dense matrix to its c x c triangular factor R, the pair (A = R^T R, b = R^T Q^T r) is
// accumulated directly.  With R = I the EKF update only depends on (A, b), and the pair is
// what a feature-sharded multi-GPU run sum-reduces over RCCL.
//   A = sum_f [ H_x^T H_x ]_f  -  sum_f G_f^T G_f
//       ^ block-sparse: per clone 6x6, 6x14; 14x14 calibration block        (k_struct_gram, VALU f64)
//                           ^ dense rank-3F downdate                          (k_syrk, v_mfma_f64_16x16x4_f64)
#include "ovp_dev.h"
#include "ovp_kernels.h"

namespace ovp {

typedef double double4_t __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// K2a: for clone slot s, X_s = rec[s] is a (2F x 21) matrix [J | C | r]; emit the packed upper triangle of
// X_s^T X_s restricted to a chunk of rows.  grid = (n_chunks, n_clones), block = 256.
// ------------------------------------------------------------------------------------------------
static constexpr int SG_TILE = 128;  // rows staged per LDS tile

__global__ __launch_bounds__(256) void k_struct_gram(const double* __restrict__ rec, int n_feats, int rows_per_chunk,
                                                      int n_chunks, double* __restrict__ gramS) {
  const int chunk = blockIdx.x, s = blockIdx.y, t = threadIdx.x;
  const int total_rows = 2 * n_feats;
  const int r0 = chunk * rows_per_chunk;
  const int r1 = min(r0 + rows_per_chunk, total_rows);
  const double* X = rec + (size_t)s * total_rows * OVP_REC;
  __shared__ double tile[SG_TILE * OVP_REC];
  // thread t < 231 owns element (p,q), p <= q, of the 21x21 Gram
  int p = 0, q = 0;
  {
    int rem = t;
    for (int pp = 0; pp < OVP_REC; ++pp) {
      const int len = OVP_REC - pp;
      if (rem < len) {
        p = pp;
        q = pp + rem;
        break;
      }
      rem -= len;
    }
  }
  double acc = 0.0;
  for (int base = r0; base < r1; base += SG_TILE) {
    const int nrows = min(SG_TILE, r1 - base);
    const double* src = X + (size_t)base * OVP_REC;
    for (int i = t; i < nrows * OVP_REC; i += 256) tile[i] = src[i];
    __syncthreads();
    if (t < OVP_GRAM_ELEMS) {
      for (int rr = 0; rr < nrows; ++rr) acc = fma(tile[rr * OVP_REC + p], tile[rr * OVP_REC + q], acc);
    }
    __syncthreads();
  }
  if (t < OVP_GRAM_ELEMS) gramS[((size_t)s * n_chunks + chunk) * OVP_GRAM_ELEMS + t] = acc;
}

// sum the row-chunk partials of K2a in a fixed order: gramR[slot][e] = sum_chunk gramS[slot][chunk][e]
__global__ __launch_bounds__(256) void k_reduce_gram(const double* __restrict__ gramS, int n_chunks,
                                                      double* __restrict__ gramR) {
  const int s = blockIdx.x, t = threadIdx.x;
  if (t >= OVP_GRAM_ELEMS) return;
  const double* src = gramS + (size_t)s * n_chunks * OVP_GRAM_ELEMS + t;
  double acc = 0.0;
  int ch = 0;
  for (; ch + 4 <= n_chunks; ch += 4) {
    const double a0 = src[(size_t)(ch + 0) * OVP_GRAM_ELEMS], a1 = src[(size_t)(ch + 1) * OVP_GRAM_ELEMS];
    const double a2 = src[(size_t)(ch + 2) * OVP_GRAM_ELEMS], a3 = src[(size_t)(ch + 3) * OVP_GRAM_ELEMS];
    acc = (((acc + a0) + a1) + a2) + a3;
  }
  for (; ch < n_chunks; ++ch) acc += src[(size_t)ch * OVP_GRAM_ELEMS];
  gramR[(size_t)s * OVP_GRAM_ELEMS + t] = acc;
}

// ------------------------------------------------------------------------------------------------
// K2b: split-K SYRK  part[split][tile] = G[rows of split]^T G[rows of split]  on the lower tile triangle.
// One workgroup (4 waves) per (tile, split); every wave runs v_mfma_f64_16x16x4_f64 over a quarter of the rows,
// the four accumulators are summed in a fixed order through LDS (deterministic).
//   A-operand lane l: A[i = l&15][k = l>>4] = G[k0+k][16*ti + i];  B-operand: B[k][j = l&15] = G[k0+k][16*tj + j]
//   C/D lane l, reg v: row = (l>>4) + 4*v, col = l&15     (f64 layout; cdna_hip_programming.md §3)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_syrk(const double* __restrict__ G, int rows, int ldg, int n_split,
                                               int rows_per_split, double* __restrict__ part) {
  const int tile = blockIdx.x, split = blockIdx.y;
  // tile -> (ti, tj), ti >= tj
  int ti = 0;
  while ((ti + 1) * (ti + 2) / 2 <= tile) ++ti;
  const int tj = tile - ti * (ti + 1) / 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int kk = lane >> 4, ij = lane & 15;
  const int k_begin = split * rows_per_split;
  const int k_end = min(k_begin + rows_per_split, rows);
  double4_t acc = {0.0, 0.0, 0.0, 0.0};
  const double* ga = G + 16 * ti + ij;
  const double* gb = G + 16 * tj + ij;
  // each wave takes rows k_begin + 4*wave + 16*step ...
  for (int k0 = k_begin + 4 * wave; k0 < k_end; k0 += 16) {
    const int row = k0 + kk;
    double av = 0.0, bv = 0.0;
    if (row < k_end) {
      av = ga[(size_t)row * ldg];
      bv = gb[(size_t)row * ldg];
    }
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
  }
  __shared__ double red[4][256];
#pragma unroll
  for (int v = 0; v < 4; ++v) red[wave][((lane >> 4) + 4 * v) * 16 + (lane & 15)] = acc[v];
  __syncthreads();
  const int t = threadIdx.x;
  const double sum = ((red[0][t] + red[1][t]) + red[2][t]) + red[3][t];
  const int ntile = gridDim.x;
  part[((size_t)split * ntile + tile) * 256 + t] = sum;
}

// ------------------------------------------------------------------------------------------------
// K2c: assemble Ab[(n+1)][lda]: rows 0..n-1 = A (full symmetric), row n = b.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int gram_index(int p, int q) {  // packed upper triangle of 21x21, p <= q
  return p * OVP_REC - (p * (p - 1)) / 2 + (q - p);
}

__global__ __launch_bounds__(256) void k_assemble(const double* __restrict__ gramS, int n_clones, int n_chunks,
                                                   const double* __restrict__ part, int n_split, int ntile,
                                                   const ColMap* __restrict__ colmap, int n, double* __restrict__ Ab,
                                                   int lda) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  const int r = blockIdx.y;  // 0..n  (n = b row)
  if (c >= n) return;
  const ColMap mc = colmap[c];
  ColMap mr;
  if (r < n) {
    mr = colmap[r];
  } else {
    mr.kind = 3;  // residual "column" 20
    mr.idx = 0;
    mr.off = 0;
    mr.pad = 0;
  }
  double s = 0.0;
  // structured part
  int slot = -1, p = -1, q = -1;  // single-slot contribution
  bool allslots = false;
  auto gcol = [](const ColMap& m) { return m.kind == 1 ? m.off : (m.kind == 2 ? 6 + m.idx : 20); };
  if (mr.kind != 0 && mc.kind != 0) {
    const int gr = gcol(mr), gc = gcol(mc);
    p = min(gr, gc);
    q = max(gr, gc);
    if (mr.kind == 1 && mc.kind == 1) {
      if (mr.idx == mc.idx) slot = mr.idx;
      else p = -1;
    } else if (mr.kind == 1) {
      slot = mr.idx;
    } else if (mc.kind == 1) {
      slot = mc.idx;
    } else {
      allslots = true;
    }
  }
  if (p >= 0) {
    const int gi = gram_index(p, q);
    if (allslots) {
#pragma unroll 8
      for (int sl = 0; sl < n_clones * n_chunks; ++sl) s += gramS[(size_t)sl * OVP_GRAM_ELEMS + gi];
    } else if (slot >= 0) {
      for (int ch = 0; ch < n_chunks; ++ch) s += gramS[((size_t)slot * n_chunks + ch) * OVP_GRAM_ELEMS + gi];
    }
  }
  // dense downdate: element (I,J) = (max,min) of (r,c) in the lower tile triangle
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
  Ab[(size_t)r * lda + c] = s;
}

// scatter a dense (cols x cols) Gram and (cols) vector given per-column state ids into Ab (zero elsewhere)
__global__ void k_scatter_gram(const double* __restrict__ Acc, const double* __restrict__ bcc, int cols,
                               const int* __restrict__ col_ids, double* __restrict__ Ab, int lda, int n) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = blockIdx.y;  // 0..cols (cols = b)
  if (j >= cols) return;
  if (i < cols)
    Ab[(size_t)col_ids[i] * lda + col_ids[j]] = Acc[(size_t)i * cols + j];
  else
    Ab[(size_t)n * lda + col_ids[j]] = bcc[j];
}

}  // namespace ovp

extern "C" {

hipError_t ovp_launch_struct_gram(const double* rec, int n_clones, int n_feats, int rows_per_chunk, int n_chunks,
                                  double* gramS, hipStream_t stream) {
  if (n_feats <= 0 || n_clones <= 0) return hipSuccess;
  hipLaunchKernelGGL(ovp::k_struct_gram, dim3(n_chunks, n_clones), dim3(256), 0, stream, rec, n_feats,
                     rows_per_chunk, n_chunks, gramS);
  return hipGetLastError();
}

hipError_t ovp_launch_reduce_gram(const double* gramS, int n_clones, int n_chunks, double* gramR, hipStream_t stream) {
  hipLaunchKernelGGL(ovp::k_reduce_gram, dim3(n_clones), dim3(256), 0, stream, gramS, n_chunks, gramR);
  return hipGetLastError();
}

hipError_t ovp_launch_syrk(const double* G, int rows, int ldg, int ncols, int n_split, double* part,
                           hipStream_t stream) {
  const int nt = (ncols + 15) / 16;
  const int ntile = nt * (nt + 1) / 2;
  int rps = (rows + n_split - 1) / n_split;
  rps = ((rps + 15) / 16) * 16;
  hipLaunchKernelGGL(ovp::k_syrk, dim3(ntile, n_split), dim3(256), 0, stream, G, rows, ldg, n_split, rps, part);
  return hipGetLastError();
}

hipError_t ovp_launch_assemble(const double* gramS, int n_clones, int n_chunks, const double* part, int n_split,
                               const ovp::ColMap* colmap, int n, double* Ab, int lda, hipStream_t stream) {
  const int nt = (n + 1 + 15) / 16;
  const int ntile = nt * (nt + 1) / 2;
  hipLaunchKernelGGL(ovp::k_assemble, dim3((n + 255) / 256, n + 1), dim3(256), 0, stream, gramS, n_clones, n_chunks,
                     part, n_split, ntile, colmap, n, Ab, lda);
  return hipGetLastError();
}

hipError_t ovp_launch_scatter_gram(const double* Acc, const double* bcc, int cols, const int* col_ids, double* Ab,
                                   int lda, int n, hipStream_t stream) {
  hipLaunchKernelGGL(ovp::k_scatter_gram, dim3((cols + 127) / 128, cols + 1), dim3(128), 0, stream, Acc, bcc, cols,
                     col_ids, Ab, lda, n);
  return hipGetLastError();
}
}
