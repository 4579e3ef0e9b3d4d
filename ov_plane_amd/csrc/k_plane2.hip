// Plane loop, second generation (update/UpdaterMSCKF.cpp:411-649): the kernels between the per-feature rows (k_plane_feat,
// k_plane.hip) and the factorization (k_chol2.hip).
//
// Every plane is a sequential EKF update, but nothing in it needs the covariance itself: with P0 = L0 L0^T the covariance at the
// start of the loop and A_j, b_j the information pairs of the planes accepted so far,
//     P_k = L0 T_k^-1 L0^T,   T_k = I + L0^T (sum_j A_j) L0 = T_(k-1) + L0^T A_k L0,
//     dx_k = L0 T_k^-1 (L0^T b_k),   b_k . dx_k = |Lt^-1 L0^T b_k|^2   (Lt Lt^T = T_k),
// so a plane costs: rows -> pair (k_gram_pair + k_plane_assemble2), two products (W = A L0, T_try = T + L0^T W, c = L0^T b),
// ONE launch of k_chol2 whose two workgroups factorize T_try (bordered with c) and the normalised Gram (bordered with its
// right-hand side, for the range part of the residual and the rank), take the gate decision, back-substitute, form dx and
// commit the state tables.  The covariance is materialised once, after the last plane.
#include "ovp_dev.h"
#include "ovp_kernels.h"
#include "k_plane2.h"

namespace ovp {

typedef double double4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int gram_index3(int p, int q) { return p * OVP_REC - (p * (p - 1)) / 2 + (q - p); }

// ------------------------------------------------------------------------------------------------------------------------
// Extended pair over [state columns | residual | plane columns] -> Ab (pair on the state columns; an out-of-state plane is
// eliminated by its 3x3 Schur complement == UpdaterHelper::nullspace_project_inplace(Hcp_big, ...), UpdaterMSCKF.cpp:603),
// the normalised, regularised Gram of the involved columns in the order `perm` (for the range part / rank), and the total
// projected residual energy.  One workgroup per row of Ab (row n = b).
// ------------------------------------------------------------------------------------------------------------------------
static constexpr int PA_THREADS = 320;  // one thread per state column (n <= 288) - no column loop, every load of a thread is issued once

__global__ __launch_bounds__(PA_THREADS) void k_plane_assemble2(const PlaneAsm a) {
  const int r = blockIdx.x;  // 0..n
  const int t = threadIdx.x;
  const int n = a.n;
  const int c = t;           // this thread's column
  __shared__ double cst[10];
  __shared__ double red[256];
  __shared__ double sh_h[PA_MAXQ][7];
  __shared__ int sh_col[PA_MAXQ][7];
  __shared__ int sh_nq;
  __shared__ double Ai[9], bc[3], xr[3], dr_sh, slam_rr, ecc[6], err_sh;
  __shared__ ColMap cm_sh[OVP_TC_MAX_TILES * 16];
  __shared__ double gsum[OVP_GRAM_ELEMS];  // Gram of the sparse rows summed over the clones

  // The kernel is a chain of memory round trips, not of arithmetic, so it is laid out by what depends on what:
  //   round 1 (addresses known up front): G^T G partials of every entry this thread needs, column classification, moments
  //   round 2 (needs the classification): the per-clone Gram entries
  //   then LDS-only steps: 3x3 inverse of the plane block, this row's diagonal, the Schur complement, normalisation.
  // Entries of the extended pair a thread needs: E(r,c), E(p0..2,c), E(c,c) and one of the 13 block-shared ones.
  // offset of entry (row, col) of the extended pair inside one split of the G^T G partials (lower tile triangle, 16 x 16 tiles)
  auto part_off = [&](int row, int col) {
    const int I = max(row, col), J = min(row, col);
    const int ti = I >> 4, tj = J >> 4;
    return (size_t)(ti * (ti + 1) / 2 + tj) * 256 + (I & 15) * 16 + (J & 15);
  };
  // the block-shared entry of this thread (if any): (row, col) in the extended index space
  int sh_row = -1, sh_col_ = -1, sh_kind = -1;
  if (!a.in_state) {
    if (t < 6) {
      const int i = t < 3 ? 0 : (t < 5 ? 1 : 2), j = t < 3 ? t : (t < 5 ? t - 2 : 2);
      sh_row = n + 1 + i, sh_col_ = n + 1 + j, sh_kind = 0;
    } else if (t >= 64 && t < 67) {
      sh_row = n, sh_col_ = n + 1 + (t - 64), sh_kind = 1;
    } else if (t >= 128 && t < 131) {
      sh_row = r, sh_col_ = n + 1 + (t - 128), sh_kind = 2;
    }
  }
  if (t == 192 && r < n) sh_row = r, sh_col_ = r, sh_kind = 3;
  const bool colv = c < n;
  const int cq = colv ? c : 0;
  // ---- round 1 ----
  // Everything this thread needs whose address is known up front is requested BEFORE anything is consumed - the first batch of
  // every sum below is the only one in the shapes that matter (30 clones x 1 chunk, <= 2 splits, <= 100 features); written as one
  // load-add loop per quantity the kernel was a chain of ~12 memory round trips (16.5 K of its 26 K cycles), the loops that remain
  // take what a first batch does not cover.  Sums are formed in the same order as before (split by split, clone by clone).
  double d_rc = 0.0, d_cc = 0.0, d_sh = 0.0;
  double d_pc[3] = {0.0, 0.0, 0.0};
  const bool ext = !a.in_state, shv = sh_kind >= 0;
  const size_t o_rc = part_off(r, cq), o_cc = part_off(cq, cq);
  const size_t o_p0 = ext ? part_off(n + 1, cq) : o_rc, o_p1 = ext ? part_off(n + 2, cq) : o_rc, o_p2 = ext ? part_off(n + 3, cq) : o_rc;
  const size_t o_sh = shv ? part_off(sh_row, sh_col_) : o_rc;
  auto part_loads = [&](int sp, double (&v)[2][6]) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const double* pb = a.part + (size_t)(sp + u < a.n_split ? sp + u : a.n_split - 1) * a.ntile * 256;
      v[u][0] = pb[o_rc];
      v[u][1] = pb[o_cc];
      v[u][2] = pb[o_p0];
      v[u][3] = pb[o_p1];
      v[u][4] = pb[o_p2];
      v[u][5] = pb[o_sh];
    }
  };
  auto part_adds = [&](int sp, const double (&v)[2][6]) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
      if (sp + u < a.n_split) {
        d_rc += v[u][0];
        d_cc += v[u][1];
        if (ext) d_pc[0] += v[u][2], d_pc[1] += v[u][3], d_pc[2] += v[u][4];
        if (shv) d_sh += v[u][5];
      }
  };
  constexpr int GB = 32;  // Gram sums: clones x chunks in flight together
  const int total = a.n_clones * a.n_chunks;
  const bool gthread = t < OVP_GRAM_ELEMS, cthread = t < 250;
  const int ce = t % 10, cpart = t / 10;
  double pv[2][6], gv[GB], cv[4];
  part_loads(0, pv);
#pragma unroll
  for (int u = 0; u < GB; ++u) gv[u] = gthread ? a.gramS[(size_t)(u < total ? u : total - 1) * OVP_GRAM_ELEMS + t] : 0.0;
#pragma unroll
  for (int u = 0; u < 4; ++u) cv[u] = (cthread && a.nf > 0) ? a.cst[(size_t)(cpart + 25 * u < a.nf ? cpart + 25 * u : 0) * 10 + ce] : 0.0;
  ColMap cmv;
  cmv.kind = 0;
  cmv.idx = cmv.off = cmv.pad = 0;
  if (t < n) cmv = a.colmap[t];
  // ---- consume ----
  part_adds(0, pv);
  for (int sp = 2; sp < a.n_split; sp += 2) {
    part_loads(sp, pv);
    part_adds(sp, pv);
  }
  if (t < n) cm_sh[t] = cmv;
  for (int i = t + PA_THREADS; i < n; i += PA_THREADS) cm_sh[i] = a.colmap[i];
  if (gthread) {
    double g = 0.0;
#pragma unroll
    for (int u = 0; u < GB; ++u)
      if (u < total) g += gv[u];
    for (int k = GB; k < total; k += GB) {
#pragma unroll
      for (int u = 0; u < GB; ++u) gv[u] = a.gramS[(size_t)(k + u < total ? k + u : total - 1) * OVP_GRAM_ELEMS + t];
#pragma unroll
      for (int u = 0; u < GB; ++u)
        if (k + u < total) g += gv[u];
    }
    gsum[t] = g;
  }
  {
    double sacc = 0.0;
    if (cthread) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (cpart + 25 * u < a.nf) sacc += cv[u];
      for (int f = cpart + 100; f < a.nf; f += 25) sacc += a.cst[(size_t)f * 10 + ce];
    }
    if (t < 256) red[t] = cthread ? sacc : 0.0;
  }
  // SLAM landmarks lying on this (out-of-state) plane: one point-on-plane row each (UpdaterMSCKF.cpp:545-552)
  if (t == 255) {
    int nq = 0;
    double rr = 0.0;
    if (!a.in_state)
      for (int q = 0; q < a.n_slam && nq < PA_MAXQ; ++q) {
        if (a.slam_plane[q] != a.plane1) continue;
        const double* pv = a.slam_p + 3 * q;
        const double* pj = a.do_fej ? a.slam_p_fej + 3 * q : pv;
        const double* cp = a.cp;
        const double* cj = a.do_fej ? a.cp_fej : a.cp;
        double d = sqrt(cp[0] * cp[0] + cp[1] * cp[1] + cp[2] * cp[2]);
        double nv[3] = {cp[0] / d, cp[1] / d, cp[2] / d};
        const double res = a.white_c * (0.0 - ((nv[0] * pv[0] + nv[1] * pv[1] + nv[2] * pv[2]) - d));
        d = sqrt(cj[0] * cj[0] + cj[1] * cj[1] + cj[2] * cj[2]);
        nv[0] = cj[0] / d;
        nv[1] = cj[1] / d;
        nv[2] = cj[2] / d;
        const double np = nv[0] * pj[0] + nv[1] * pj[1] + nv[2] * pj[2];
        for (int k = 0; k < 3; ++k) {
          sh_h[nq][k] = a.white_c * nv[k];
          sh_h[nq][3 + k] = a.white_c * 1.0 / d * (pj[k] - np * nv[k] - d * nv[k]);
          sh_col[nq][k] = a.slam_id[q] + k;
          sh_col[nq][3 + k] = n + 1 + k;
        }
        sh_h[nq][6] = res;
        sh_col[nq][6] = n;
        rr += res * res;
        ++nq;
      }
    sh_nq = nq;
    slam_rr = rr;
  }
  __syncthreads();
  if (t < 10) {
    double acc = 0.0;
    for (int l = 0; l < 25; ++l) acc += red[l * 10 + t];
    cst[t] = acc;
  }
  auto classify = [&](int col) {
    ColMap m;
    m.kind = 0;
    m.idx = m.off = m.pad = 0;
    if (col < n) {
      m = cm_sh[col];
      if (a.plane_sid >= 0 && col >= a.plane_sid && col < a.plane_sid + 3) {
        m.kind = 4;
        m.idx = col - a.plane_sid;
      }
    } else if (col == n) {
      m.kind = 3;
    } else if (a.plane_sid < 0) {
      m.kind = 4;
      m.idx = col - n - 1;
    }
    return m;
  };
  // ---- round 2: the per-clone Gram entry of every needed (row, col); plane / clone-sum cases load a dummy element ----
  auto gram_load = [&](int row, int col) {
    const ColMap mr = classify(row), mc = classify(col);
    const int gr = mr.kind == 1 ? mr.off : (mr.kind == 2 ? 6 + mr.idx : 20);
    const int gc = mc.kind == 1 ? mc.off : (mc.kind == 2 ? 6 + mc.idx : 20);
    const int gi = gram_index3(min(gr, gc), max(gr, gc));
    const int slot = mr.kind == 1 ? mr.idx : (mc.kind == 1 ? mc.idx : 0);
    double g1 = 0.0;
    for (int ch = 0; ch < a.n_chunks; ++ch) g1 += a.gramS[((size_t)slot * a.n_chunks + ch) * OVP_GRAM_ELEMS + gi];
    return g1;
  };
  const double g_rc = gram_load(r, cq), g_cc = gram_load(cq, cq);
  double g_pc[3] = {0.0, 0.0, 0.0};
  if (!a.in_state)
    for (int k = 0; k < 3; ++k) g_pc[k] = gram_load(n + 1 + k, cq);
  const double g_sh = sh_kind >= 0 ? gram_load(sh_row, sh_col_) : 0.0;
  __syncthreads();  // cst, gsum, SLAM rows
  // ---- LDS-only from here: combine the terms of an entry ----
  auto finish = [&](int row, int col, double g1, double d) {
    const ColMap mr = classify(row), mc = classify(col);
    const bool pl = (mr.kind == 4 || mc.kind == 4);
    double s_pl = 0.0;
    {
      const int i = min(mr.idx, mc.idx), j = max(mr.idx, mc.idx);
      const double hh = cst[(mr.kind == 4 && mc.kind == 4) ? (i == 0 ? j : (i == 1 ? 2 + j : 5)) : 0];
      const double hr = cst[6 + ((mr.kind == 4 ? mr.idx : mc.idx) % 3)];
      if (mr.kind == 4 && mc.kind == 4) s_pl = hh;
      else if (mr.kind == 3 || mc.kind == 3) s_pl = hr;
    }
    const int gr = mr.kind == 1 ? mr.off : (mr.kind == 2 ? 6 + mr.idx : 20);
    const int gc = mc.kind == 1 ? mc.off : (mc.kind == 2 ? 6 + mc.idx : 20);
    const int gi = gram_index3(min(gr, gc), max(gr, gc));
    const bool any1 = (mr.kind == 1 || mc.kind == 1);
    const bool both1_diff = (mr.kind == 1 && mc.kind == 1 && mr.idx != mc.idx);
    double s_g = 0.0;
    if (!pl && mr.kind != 0 && mc.kind != 0) s_g = any1 ? (both1_diff ? 0.0 : g1) : gsum[gi];
    double s = (pl ? s_pl : s_g) - d;
    for (int q = 0; q < sh_nq; ++q) {
      double hr = 0.0, hc = 0.0;
#pragma unroll
      for (int k = 0; k < 7; ++k) {
        if (sh_col[q][k] == row) hr = sh_h[q][k];
        if (sh_col[q][k] == col) hc = sh_h[q][k];
      }
      if (!(row == n && col == n)) s += hr * hc;
    }
    return s;
  };
  if (sh_kind >= 0) {
    const double v = finish(sh_row, sh_col_, g_sh, d_sh);
    if (sh_kind == 0) ecc[t] = v;
    else if (sh_kind == 1) bc[t - 64] = v;
    else if (sh_kind == 2) xr[t - 128] = v;
    else err_sh = v;
  } else if (a.in_state && t < 3) {
    bc[t] = xr[t] = 0.0;
  }
  if (t == 192 && r >= n) err_sh = 0.0;
  __syncthreads();
  if (t == 0) {
    if (!a.in_state) {
      const double a00 = ecc[0], a01 = ecc[1], a02 = ecc[2], a11 = ecc[3], a12 = ecc[4], a22 = ecc[5];
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
    } else {
      for (int k = 0; k < 9; ++k) Ai[k] = 0.0;
    }
  }
  __syncthreads();
  auto schur = [&](const double (&u)[3], const double (&v)[3]) {
    const double t0 = Ai[0] * v[0] + Ai[1] * v[1] + Ai[2] * v[2];
    const double t1 = Ai[3] * v[0] + Ai[4] * v[1] + Ai[5] * v[2];
    const double t2 = Ai[6] * v[0] + Ai[7] * v[1] + Ai[8] * v[2];
    return u[0] * t0 + u[1] * t1 + u[2] * t2;
  };
  const double xrr[3] = {xr[0], xr[1], xr[2]};
  if (t == 192) dr_sh = r < n ? err_sh - (a.in_state ? 0.0 : schur(xrr, xrr)) : 0.0;
  if (r == n && t == 193) {
    const double bcc[3] = {bc[0], bc[1], bc[2]};
    a.scal[0] = cst[9] + slam_rr - (a.in_state ? 0.0 : schur(bcc, bcc));
  }
  __syncthreads();
  if (!colv) return;
  const double dr = dr_sh;
  const int pr = r < n ? a.perm[r] : -1;
  double xc[3] = {0.0, 0.0, 0.0};
  if (!a.in_state)
    for (int k = 0; k < 3; ++k) xc[k] = finish(n + 1 + k, c, g_pc[k], d_pc[k]);
  const double v = finish(r, c, g_rc, d_rc) - (a.in_state ? 0.0 : schur(xrr, xc));
  a.Ab[(size_t)r * a.lda + c] = v;
  const int pc = a.perm[c];
  if (pc >= 0 && (pr >= 0 || r == n)) {
    const double dc = (r == c) ? dr : finish(c, c, g_cc, d_cc) - (a.in_state ? 0.0 : schur(xc, xc));
    if (r == n) {
      a.bn[pc] = dc > 0.0 ? v / sqrt(dc) : 0.0;
    } else {
      double w;
      if (dr > 0.0 && dc > 0.0) w = (r == c) ? 1.0 + a.eps : v / sqrt(dr * dc);
      else w = (r == c) ? 1.0 : 0.0;
      a.An[(size_t)pr * a.ldn + pc] = w;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// T_try = T_cur + L0^T W on the lower tile triangle (W = A L0 from k_gemm4), c = L0^T b.  T_cur / T_try are the two halves of
// Tbuf selected by the device word *cur (toggled by k_chol2 when a plane is accepted).  L0 is lower triangular: the products
// start at k = 16 bi.  grid = (nt, nt + 1): block row nt computes c.
// ------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_plane_dT(int n, const double* __restrict__ L0, int ld, const double* __restrict__ W,
                                                   const double* __restrict__ b, double* __restrict__ Tbuf, size_t tstride,
                                                   const int* __restrict__ cur, double* __restrict__ crow) {
  const int bi = blockIdx.y, bj = blockIdx.x;
  const int nt = gridDim.x;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int lr = lane >> 4, lc = lane & 15;
  __shared__ double red[4][256];
  if (bi == nt) {
    // c[j] = sum_K L0[K][j] b[K], j in tile bj: 16 columns x 16 partial sums
    const int j = 16 * bj + (tid & 15), p = tid >> 4;
    double s = 0.0;
    if (j < n)
      for (int K = 16 * bj + p; K < n; K += 16) s = fma(L0[(size_t)K * ld + j], b[K], s);
    red[0][tid] = s;
    __syncthreads();
    if (tid < 16) {
      double acc = 0.0;
      for (int q = 0; q < 16; ++q) acc += red[0][q * 16 + tid];
      if (16 * bj + tid < n) crow[16 * bj + tid] = acc;
    }
    return;
  }
  if (bj > bi) return;
  const int sel = *cur;
  const double* Tc = Tbuf + (size_t)sel * tstride;
  double* Tt = Tbuf + (size_t)(sel ^ 1) * tstride;
  const int i0 = bi * 16, j0 = bj * 16;
  const int ai = i0 + lc, bjj = j0 + lc;
  const int ks0 = (16 * bi) >> 2;  // first k-step (4 consecutive k) with a non-zero L0[k][i0..]
  const int nsteps = (n + 3) >> 2;
  double4_t acc = {0.0, 0.0, 0.0, 0.0};
  for (int st = ks0 + wave; st < nsteps; st += 16) {
    double av[4], bv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = 4 * (st + 4 * u) + lr;
      av[u] = 0.0;
      bv[u] = 0.0;
      if (k < n && (st + 4 * u) < nsteps) {
        if (ai < n) av[u] = L0[(size_t)k * ld + ai];
        if (bjj < n) bv[u] = W[(size_t)k * ld + bjj];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
  }
#pragma unroll
  for (int v = 0; v < 4; ++v) red[wave][(lr + 4 * v) * 16 + lc] = acc[v];
  __syncthreads();
  const int row = tid >> 4, col = tid & 15;
  const double sum = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
  const int gr = i0 + row, gc = j0 + col;
  if (gr < n && gc < n) Tt[(size_t)gr * ld + gc] = Tc[(size_t)gr * ld + gc] + sum;
}

// One launch instead of a fill per buffer at the start of a plane loop (each is a kernel of its own on this stack, ~4 us of
// boundary): up to eight regions, sizes in bytes (multiples of 4); region r is cleared by the blocks b = r, r + 8, r + 16, ...
// Two regions may ask for an identity pattern instead of zeros (doubles): 1 = [k][16][16] blocks with a unit diagonal (the inverted
// diagonal blocks k_fwdsub reads), 2 = the tile-packed lower triangle of an n x n matrix with identity diagonal tiles (its packed
// factor) - what a plane of the loop that factorized only a leading block leaves untouched behind that block.
struct ZeroJob {
  void* ptr[8];
  unsigned long long bytes[8];
  int pattern[8];
  int ntn;  // tile rows of the packed factor (pattern 2)
};
__global__ __launch_bounds__(256) void k_zero_regions(const ZeroJob z) {
  const int r = blockIdx.x % 8, b = blockIdx.x / 8, nb = (gridDim.x + 7 - r) / 8;
  const unsigned long long words = z.bytes[r] >> 2;
  if (z.pattern[r] == 0) {
    unsigned int* p = reinterpret_cast<unsigned int*>(z.ptr[r]);
    for (unsigned long long i = (unsigned long long)b * 256 + threadIdx.x; i < words; i += (unsigned long long)nb * 256) p[i] = 0u;
    return;
  }
  double* p = reinterpret_cast<double*>(z.ptr[r]);
  const unsigned long long dbl = words >> 1;
  for (unsigned long long i = (unsigned long long)b * 256 + threadIdx.x; i < dbl; i += (unsigned long long)nb * 256) {
    const int e = (int)(i & 255), tile = (int)(i >> 8);
    bool diag_tile = true;
    if (z.pattern[r] == 2) {  // tile (i, j) of the packed triangle sits at j * ntn - j (j - 1) / 2 + (i - j)
      diag_tile = false;
      for (int j = 0; j < z.ntn; ++j) diag_tile |= (tile == j * z.ntn - (j * (j - 1)) / 2);
    }
    p[i] = (diag_tile && (e >> 4) == (e & 15)) ? 1.0 : 0.0;
  }
}

// dst <- the half of buf selected by *cur (lower triangle mirrored to a full symmetric matrix when `sym`)
__global__ __launch_bounds__(256) void k_select_copy(double* __restrict__ dst, const double* __restrict__ buf, size_t stride,
                                                      const int* __restrict__ cur, int n, int ld, int sym) {
  const double* src = buf + (size_t)(*cur) * stride;
  const int r = blockIdx.x;
  for (int c = threadIdx.x; c < n; c += 256) {
    const double v = (sym && c > r) ? src[(size_t)c * ld + r] : src[(size_t)r * ld + c];
    dst[(size_t)r * ld + c] = v;
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// Plane loop on a sub-state (state above the tile factorization's limit, ovp_api_plane.hip: plane_update_substate).  The loop runs on the
// ns involved columns s; the correction of the WHOLE state for an accepted plane k is dx_k = P0[:, s] u_k with
//     u_k = (I + A^(k) P0ss)^-1 b_k = b_k - A^(k) dx_k[s],        A^(k) = sum of the pairs accepted so far including plane k,
// (push-through identity; no solve with P0ss).  One block per row r of the pair: Asum[r, :] += A_k[r, :], u[r] = b_k[r] - Asum[r, :] dx.
// A rejected plane (res[1] == 0) leaves Asum alone and gets u = 0.
// ------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_plane_sub_accum(const double* __restrict__ res, const double* __restrict__ Ab,
                                                          double* __restrict__ Asum, const double* __restrict__ dx,
                                                          double* __restrict__ u, int ns, int ld) {
  const int r = blockIdx.x, t = threadIdx.x;
  __shared__ double red[256];
  if (!(res[1] > 0.5)) {
    if (t == 0) u[r] = 0.0;
    return;
  }
  double s = 0.0;
  for (int c = t; c < ns; c += 256) {
    const double a = Asum[(size_t)r * ld + c] + Ab[(size_t)r * ld + c];
    Asum[(size_t)r * ld + c] = a;
    s = fma(a, dx[c], s);
  }
  red[t] = s;
  __syncthreads();
  for (int w = 128; w >= 1; w >>= 1) {
    if (t < w) red[t] += red[t + w];
    __syncthreads();
  }
  if (t == 0) u[r] = Ab[(size_t)ns * ld + r] - red[0];
}

}  // namespace ovp

extern "C" {
hipError_t ovp_launch_plane_sub_accum(const double* res, const double* Ab, double* Asum, const double* dx, double* u, int ns, int ld,
                                      hipStream_t stream) {
  hipLaunchKernelGGL(ovp::k_plane_sub_accum, dim3(ns), dim3(256), 0, stream, res, Ab, Asum, dx, u, ns, ld);
  return hipGetLastError();
}
hipError_t ovp_launch_plane_assemble2(const ovp::PlaneAsm* a, hipStream_t stream) {
  hipLaunchKernelGGL(ovp::k_plane_assemble2, dim3(a->n + 1), dim3(ovp::PA_THREADS), 0, stream, *a);
  return hipGetLastError();
}
hipError_t ovp_launch_plane_dT(int n, const double* L0, int ld, const double* W, const double* b, double* Tbuf, size_t tstride,
                               const int* cur, double* crow, hipStream_t stream) {
  const int nt = (n + 15) / 16;
  hipLaunchKernelGGL(ovp::k_plane_dT, dim3(nt, nt + 1), dim3(256), 0, stream, n, L0, ld, W, b, Tbuf, tstride, cur, crow);
  return hipGetLastError();
}
hipError_t ovp_launch_zero_regions(void* const* ptr, const size_t* bytes, int count, hipStream_t stream) {
  return ovp_launch_fill_regions(ptr, bytes, nullptr, count, 0, stream);
}
hipError_t ovp_launch_fill_regions(void* const* ptr, const size_t* bytes, const int* pattern, int count, int ntn, hipStream_t stream) {
  ovp::ZeroJob z;
  size_t most = 0;
  if (count > 8) return hipErrorInvalidValue;
  for (int i = 0; i < 8; ++i) {
    z.ptr[i] = i < count ? ptr[i] : nullptr;
    z.bytes[i] = i < count ? (unsigned long long)bytes[i] : 0ull;
    z.pattern[i] = (i < count && pattern) ? pattern[i] : 0;
    if (i < count && bytes[i] > most) most = bytes[i];
  }
  z.ntn = ntn;
  int per = (int)((most / 4 + 4 * 256 - 1) / (4 * 256));  // a thread of the largest region clears ~4 words
  if (per < 1) per = 1;
  if (per > 512) per = 512;
  hipLaunchKernelGGL(ovp::k_zero_regions, dim3(8 * per), dim3(256), 0, stream, z);
  return hipGetLastError();
}
hipError_t ovp_launch_select_copy(double* dst, const double* buf, size_t stride, const int* cur, int n, int ld, int sym,
                                  hipStream_t stream) {
  hipLaunchKernelGGL(ovp::k_select_copy, dim3(n), dim3(256), 0, stream, dst, buf, stride, cur, n, ld, sym);
  return hipGetLastError();
}
}
