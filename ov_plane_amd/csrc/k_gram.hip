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
#include <cstdlib>

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
// K2ab: both Gram products in ONE launch, both on v_mfma_f64_16x16x4_f64 (k_struct_gram / k_syrk above are the first
// versions, kept for the plane path's small batches and as cross-checks).
//   blocks [0, n_struct):   (chunk, slot) of the structured part: X = rec[slot] rows of the chunk, 21 columns padded to two
//                           16-column groups, three tiles of X^T X, emitted in the packed 231-element layout of k_struct_gram
//   blocks [n_struct, ..):  (macro tile, split) of the dense downdate G^T G: a 64 x 64 macro tile = 4 x 4 MFMA tiles held in
//                           registers by every wave (one operand load feeds four MFMAs: 8 loads per 16 MFMAs instead of the
//                           2 per MFMA that made k_syrk L2-bandwidth bound), the four waves split the rows of the block and
//                           are summed through LDS in a fixed order.
// A 16 x 16 x 4 f64 MFMA occupies its SIMD for 64 cycles, so the whole product is ~10 K cycles per SIMD when spread evenly:
// the launch is sized for one or two blocks per CU.
// ------------------------------------------------------------------------------------------------
struct GramPairJob {
  const double* rec;
  int n_feats, n_clones, rows_per_chunk, n_chunks;
  double* gramS;
  const double* G;
  int rows, ldg, nt16, n_macro_rows, n_split, rows_per_split, ntile;
  double* part;
};

__global__ __launch_bounds__(256) void k_gram_pair(const GramPairJob j) {
  __shared__ __attribute__((aligned(16))) double red[4][4][256];  // [wave][tile of the round][element]  32 KB
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int kk = lane >> 4, ij = lane & 15;
  const int n_struct = j.n_chunks * j.n_clones;
  if ((int)blockIdx.x < n_struct) {
    const int slot = blockIdx.x / j.n_chunks, chunk = blockIdx.x - slot * j.n_chunks;
    const int total_rows = 2 * j.n_feats;
    const int r0 = chunk * j.rows_per_chunk;
    const int r1 = min(r0 + j.rows_per_chunk, total_rows);
    const double* X = j.rec + (size_t)slot * total_rows * OVP_REC;
    const bool hi_ok = 16 + ij < OVP_REC;  // second column group: columns 16..20
    double4_t a00 = {0.0, 0.0, 0.0, 0.0}, a10 = a00, a11 = a00;
    // raw loads from clamped addresses; rows past the chunk and the padding columns are zeroed when the operand is USED
    // (a select right behind the load would make every load a synchronous one)
    auto load = [&](int k0, double& x0, double& x1) {
      const int row = k0 + kk;
      const double* src = X + (size_t)(row < r1 ? row : r0) * OVP_REC + ij;
      x0 = src[0];
      x1 = src[hi_ok ? 16 : 0];
    };
    // all of the wave's rows are requested before the first MFMA (a step is only 3 MFMAs, far less than a memory round
    // trip): SG_STEPS steps of 4 rows per wave and pass
    constexpr int SG_STEPS = 8;
    for (int kb = r0 + 4 * wave; kb < r1; kb += 16 * SG_STEPS) {
      double x0[SG_STEPS], x1[SG_STEPS];
#pragma unroll
      for (int d = 0; d < SG_STEPS; ++d) load(kb + 16 * d, x0[d], x1[d]);
#pragma unroll
      for (int d = 0; d < SG_STEPS; ++d) {
        if (kb + 16 * d < r1) {
          const bool ok = kb + 16 * d + kk < r1;
          const double u0 = ok ? x0[d] : 0.0;
          const double u1 = (ok && hi_ok) ? x1[d] : 0.0;
          a00 = __builtin_amdgcn_mfma_f64_16x16x4f64(u0, u0, a00, 0, 0, 0);
          a10 = __builtin_amdgcn_mfma_f64_16x16x4f64(u1, u0, a10, 0, 0, 0);
          a11 = __builtin_amdgcn_mfma_f64_16x16x4f64(u1, u1, a11, 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int e = (kk + 4 * v) * 16 + ij;
      red[wave][0][e] = a00[v];
      red[wave][1][e] = a10[v];
      red[wave][2][e] = a11[v];
    }
    __syncthreads();
    if (t < OVP_GRAM_ELEMS) {
      int p = 0, q = 0, rem = t;
      for (int pp = 0; pp < OVP_REC; ++pp) {
        const int len = OVP_REC - pp;
        if (rem < len) {
          p = pp;
          q = pp + rem;
          break;
        }
        rem -= len;
      }
      // element (p, q), p <= q, = entry (row q, column p) of the lower tile triangle
      const int tl = (q >> 4) + (p >> 4);  // (0,0) -> 0, (1,0) -> 1, (1,1) -> 2
      const int e = (q & 15) * 16 + (p & 15);
      j.gramS[((size_t)slot * j.n_chunks + chunk) * OVP_GRAM_ELEMS + t] =
          ((red[0][tl][e] + red[1][tl][e]) + red[2][tl][e]) + red[3][tl][e];
    }
    return;
  }
  // ---- dense part ----
  const int b = blockIdx.x - n_struct;
  const int macro = b / j.n_split, split = b - macro * j.n_split;
  int mi = 0;
  while ((mi + 1) * (mi + 2) / 2 <= macro) ++mi;
  const int mj = macro - mi * (mi + 1) / 2;
  const int k_begin = split * j.rows_per_split;
  const int k_end = min(k_begin + j.rows_per_split, j.rows);
  const bool diag = (mi == mj);
  // live tiles (wave-uniform): tile row a = 4 mi + ia, tile column bb = 4 mj + ib, both < nt16, a >= bb
  auto live = [&](int ia, int ib) { return 4 * mi + ia < j.nt16 && 4 * mj + ib < j.nt16 && (!diag || ia >= ib); };
  double4_t acc[4][4];
#pragma unroll
  for (int ia = 0; ia < 4; ++ia)
#pragma unroll
    for (int ib = 0; ib < 4; ++ib) acc[ia][ib] = double4_t{0.0, 0.0, 0.0, 0.0};
  const int cI = 64 * mi + ij, cJ = 64 * mj + ij;
  // raw loads from clamped addresses (live tiles only touch columns < 16 nt16 <= ldg); rows past the block are zeroed in
  // the B operand when it is used - a select right behind the load would make every load a synchronous one
  auto load = [&](int k0, double (&av)[4], double (&bv)[4]) {
    const int row = k0 + kk;
    const double* src = j.G + (size_t)(row < k_end ? row : k_begin) * j.ldg;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ca = cI + 16 * q, cb = cJ + 16 * q;
      av[q] = src[ca < j.ldg ? ca : 0];
      bv[q] = src[cb < j.ldg ? cb : 0];
    }
  };
  // operand ring: the loads of DEPTH steps are in flight while one step's sixteen MFMAs execute
  constexpr int DEPTH = 4;
  double ra[DEPTH][4], rb[DEPTH][4];
  int k0 = k_begin + 4 * wave;
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) load(k0 + 16 * d, ra[d], rb[d]);
  for (; k0 < k_end; k0 += 16 * DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      if (k0 + 16 * d < k_end) {
        const bool ok = k0 + 16 * d + kk < k_end;
        double bz[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) bz[q] = ok ? rb[d][q] : 0.0;
#pragma unroll
        for (int ia = 0; ia < 4; ++ia)
#pragma unroll
          for (int ib = 0; ib < 4; ++ib)
            if (live(ia, ib))
              acc[ia][ib] = __builtin_amdgcn_mfma_f64_16x16x4f64(ra[d][ia], bz[ib], acc[ia][ib], 0, 0, 0);
      }
      load(k0 + 16 * (d + DEPTH), ra[d], rb[d]);
    }
  }
  // four rounds of four tiles: waves -> LDS -> fixed-order sum -> part[split][tile][256]
#pragma unroll
  for (int ia = 0; ia < 4; ++ia) {
#pragma unroll
    for (int ib = 0; ib < 4; ++ib) {
      if (live(ia, ib)) {
#pragma unroll
        for (int v = 0; v < 4; ++v) red[wave][ib][(kk + 4 * v) * 16 + ij] = acc[ia][ib][v];
      }
    }
    __syncthreads();
#pragma unroll
    for (int ib = 0; ib < 4; ++ib) {
      if (live(ia, ib)) {
        const int a = 4 * mi + ia, bb = 4 * mj + ib;
        const int tile = a * (a + 1) / 2 + bb;
        j.part[((size_t)split * j.ntile + tile) * 256 + t] =
            ((red[0][ib][t] + red[1][ib][t]) + red[2][ib][t]) + red[3][ib][t];
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// K2c: assemble Ab[(n+1)][lda]: rows 0..n-1 = A (full symmetric), row n = b.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int gram_index(int p, int q) {  // packed upper triangle of 21x21, p <= q
  return p * OVP_REC - (p * (p - 1)) / 2 + (q - p);
}

// One block per 16 x 16 tile of the LOWER tile triangle of the (n + 1) x (n + 1) pair (row n = b): the partials of a tile are
// 2 KB contiguous per split (until round 4 a block owned a ROW of Ab and looked every element up at (max, min): the upper half
// of a row then reads one double per 128-byte line - 22.7 MB fetched for 0.45 MB of output, `profiles/r04_c_hbm_traffic_pmc.json`),
// the mirror image of an off-diagonal tile goes out through an LDS transpose.  Same sums in the same order as before.
__global__ __launch_bounds__(256) void k_assemble(const double* __restrict__ gramS, int n_clones, int n_chunks,
                                                   const double* __restrict__ part, int n_split, int ntile,
                                                   const ColMap* __restrict__ colmap, int n, double* __restrict__ Ab,
                                                   int lda) {
  __shared__ double tr[16][17];
  // tile (ti, tj), ti >= tj, from the list index ti (ti + 1) / 2 + tj
  int ti = 0;
  {
    const int b = blockIdx.x;
    while ((ti + 1) * (ti + 2) / 2 <= b) ++ti;
  }
  const int tj = blockIdx.x - ti * (ti + 1) / 2;
  const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
  const int r = 16 * ti + ty, c = 16 * tj + tx;  // element (r, c) of the pair; r <= n is a row of Ab, c < n a column
  const bool live = r <= n && c < n;
  // everything whose address is known is requested before anything is consumed; sums in a fixed order (split by split, slot by slot)
  constexpr int NB = 32;
  double dv[NB];
  const int I = max(r, c), J = min(r, c);  // (a diagonal tile holds its lower half)
  const size_t st = (size_t)ntile * 256;
  const double* pp = part + (size_t)blockIdx.x * 256 + (I & 15) * 16 + (J & 15);
#pragma unroll
  for (int u = 0; u < NB; ++u) dv[u] = pp[(size_t)(u < n_split ? u : n_split - 1) * st];
  ColMap mc, mr;
  mc.kind = mr.kind = 0;
  mc.idx = mc.off = mc.pad = mr.idx = mr.off = mr.pad = 0;
  if (c < n) mc = colmap[c];
  if (r < n) mr = colmap[r];
  else if (r == n) mr.kind = 3;  // residual "column" 20
  double s = 0.0;
  // structured part
  int slot = -1, p = -1, q = -1;  // single-slot contribution
  bool allslots = false;
  auto gcol = [](const ColMap& m) { return m.kind == 1 ? m.off : (m.kind == 2 ? 6 + m.idx : 20); };
  if (live && mr.kind != 0 && mc.kind != 0) {
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
    // one list for both cases: entries first .. first + count - 1 of gramS (all slots and chunks, or the chunks of one slot)
    const int first = allslots ? 0 : slot * n_chunks;
    const int count = allslots ? n_clones * n_chunks : (slot >= 0 ? n_chunks : 0);
    if (count > 0) {
      double gv[NB];
#pragma unroll
      for (int u = 0; u < NB; ++u) gv[u] = gramS[(size_t)(first + (u < count ? u : count - 1)) * OVP_GRAM_ELEMS + gi];
#pragma unroll
      for (int u = 0; u < NB; ++u)
        if (u < count) s += gv[u];
      for (int sl = NB; sl < count; ++sl) s += gramS[(size_t)(first + sl) * OVP_GRAM_ELEMS + gi];
    }
  }
  {
    double d = 0.0;
#pragma unroll
    for (int u = 0; u < NB; ++u)
      if (u < n_split) d += dv[u];
    for (int sp = NB; sp < n_split; ++sp) d += pp[(size_t)sp * st];
    s -= d;
  }
  if (live) Ab[(size_t)r * lda + c] = s;
  if (ti != tj) {  // the mirror image: element (c, r) = row 16 tj + .., column 16 ti + .. (a column of Ab: < n)
    tr[ty][tx] = s;
    __syncthreads();
    const int rm = 16 * tj + ty, cm = 16 * ti + tx;
    if (rm < n && cm < n) Ab[(size_t)rm * lda + cm] = tr[tx][ty];
  }
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

// the same with += : a second pair (the dense blocks of ovp_msckf_dense_blocks) joins the one K2 assembled.  One thread per entry,
// every entry of Ab is touched by at most one thread (col_ids are distinct)
__global__ void k_scatter_gram_add(const double* __restrict__ Acc, const double* __restrict__ bcc, int cols,
                                   const int* __restrict__ col_ids, double* __restrict__ Ab, int lda, int n) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = blockIdx.y;  // 0..cols (cols = b)
  if (j >= cols) return;
  if (i < cols)
    Ab[(size_t)col_ids[i] * lda + col_ids[j]] += Acc[(size_t)i * cols + j];
  else
    Ab[(size_t)n * lda + col_ids[j]] += bcc[j];
}

}  // namespace ovp

extern "C" {

hipError_t ovp_launch_scatter_gram_add(const double* Acc, const double* bcc, int cols, const int* col_ids, double* Ab, int lda, int n,
                                       hipStream_t stream) {
  hipLaunchKernelGGL(ovp::k_scatter_gram_add, dim3((cols + 127) / 128, cols + 1), dim3(128), 0, stream, Acc, bcc, cols, col_ids, Ab,
                     lda, n);
  return hipGetLastError();
}

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

// both Gram products of K2 in one launch; *n_split_used tells the assemble kernel how many partials to sum
hipError_t ovp_launch_gram_pair(const double* rec, int n_clones, int n_feats, int rows_per_chunk, int n_chunks,
                                double* gramS, const double* G, int rows, int ldg, int ncols, int n_split_cap,
                                double* part, int* n_split_used, hipStream_t stream) {
  ovp::GramPairJob j;
  j.rec = rec;
  j.n_feats = n_feats;
  j.n_clones = n_clones;
  j.rows_per_chunk = rows_per_chunk;
  j.n_chunks = n_chunks;
  j.gramS = gramS;
  j.G = G;
  j.rows = rows;
  j.ldg = ldg;
  j.nt16 = (ncols + 15) / 16;
  if (16 * j.nt16 > ldg) return hipErrorInvalidValue;
  j.n_macro_rows = (j.nt16 + 3) / 4;
  j.ntile = j.nt16 * (j.nt16 + 1) / 2;
  const int n_macro = j.n_macro_rows * (j.n_macro_rows + 1) / 2;
  int ns = (256 + n_macro / 2) / n_macro;  // about one dense block per CU
  const int by_rows = rows / 64;           // at least four steps per wave
  if (ns > by_rows) ns = by_rows;
  if (ns > n_split_cap) ns = n_split_cap;
  if (ns < 1) ns = 1;
  int rps = (rows + ns - 1) / ns;
  rps = ((rps + 15) / 16) * 16;
  ns = (rows + rps - 1) / rps;
  j.n_split = ns;
  j.rows_per_split = rps;
  j.part = part;
  *n_split_used = ns;
  hipLaunchKernelGGL(ovp::k_gram_pair, dim3(n_chunks * n_clones + n_macro * ns), dim3(256), 0, stream, j);
  return hipGetLastError();
}

hipError_t ovp_launch_assemble(const double* gramS, int n_clones, int n_chunks, const double* part, int n_split,
                               const ovp::ColMap* colmap, int n, double* Ab, int lda, hipStream_t stream) {
  const int nt = (n + 1 + 15) / 16;
  const int ntile = nt * (nt + 1) / 2;
  hipLaunchKernelGGL(ovp::k_assemble, dim3(ntile), dim3(256), 0, stream, gramS, n_clones, n_chunks, part, n_split, ntile, colmap, n,
                     Ab, lda);
  return hipGetLastError();
}

hipError_t ovp_launch_scatter_gram(const double* Acc, const double* bcc, int cols, const int* col_ids, double* Ab,
                                   int lda, int n, hipStream_t stream) {
  hipLaunchKernelGGL(ovp::k_scatter_gram, dim3((cols + 127) / 128, cols + 1), dim3(128), 0, stream, Acc, bcc, cols,
                     col_ids, Ab, lda, n);
  return hipGetLastError();
}
}
