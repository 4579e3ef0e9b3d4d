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
#include "k_tile_body.h"
#include "k_dpp.h"
#include <utility>
#include <cstdlib>

namespace ovp {

static constexpr int NR = 64;               // rows handled by one wave (2 * OVP_MAX_MEAS)
static constexpr int LCOLS = OVP_BSCR;          // packed column-major lower triangle with even row starts
__device__ __forceinline__ int coff(int k) {  // offset of column k; element (i,k) lives at coff(k) + i - (k & ~1)
  const int pr = k >> 1;
  return 130 * pr - 2 * pr * pr + (k & 1) * (64 - 2 * pr);
}

// per-feature scratch for B in device memory: the two columns of an observation interleaved, element (i, 2 p + rr), i >= 2 p, at
// boff(p) + 2 (i - 2 p) + rr - same footprint as the LDS layout above, but a lane's two columns are 16 contiguous bytes
__device__ __forceinline__ int boff(int p) { return 130 * p - 2 * p * p; }

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

typedef double double2_t __attribute__((ext_vector_type(2)));

// The 16 columns of a block of the bordered factorization, all rows at once (k_chol2.hip: fused_elim16): d = row r = lane & 15 of
// the diagonal tile, a copy in every 16-lane DPP row; p = this lane's row of the matrix (DPP row g = row tile g), the block's 16
// entries.  Columns c < ncol are pivots: both are scaled by 1 / sqrt(pivot) (v_rsq_f64 + one Newton step folded into the scaling)
// and the later columns updated with the column of L broadcast INSIDE the FMA (v_fmac_f64_dpp row_newbcast) - no LDS round trip and
// no VALU -> SGPR -> VALU hop per column, which is what the column loop this replaces consisted of (a v_readlane pair for the
// pivot, another for the next column's multiplier, the column exchanged through LDS).  Columns >= ncol (the corner of the border)
// are only updated.  A row tile above the diagonal one holds zeros and keeps them.
__device__ __forceinline__ void k1_elim16(double (&d)[16], double (&p)[16], const int ncol, bool& spd) {
  double piv = bcast_row<0>(d[0]);
  static_for<16>([&](auto cc) {
    constexpr int c = decltype(cc)::value;
    if (c < ncol) {  // (wave-uniform)
      spd = spd && (piv > 0.0);
      const double y0 = __builtin_amdgcn_rsq(piv);
      const double hy = (0.5 * piv) * y0;
      const double ly = d[c] * y0, py = p[c] * y0;
      const double e = fma(-hy, y0, 0.5);
      const double l = fma(ly, e, ly);
      const double q = fma(py, e, py);
      double nl;  // -l, the DPP operand of the updates: its two wait states go with the write (k_dpp.h)
      asm("v_fma_f64 %0, -%1, %2, -%1\n\ts_nop 1" : "=v"(nl) : "v"(ly), "v"(e));
      d[c] = l;
      p[c] = q;
      if constexpr (c + 1 < 16) {
        fmac_bcast<c + 1>(d[c + 1], nl, l);
        piv = bcast_row<c + 1>(d[c + 1]);
        fmac_bcast<c + 1>(p[c + 1], nl, q);
        static_for<14 - c>([&](auto jc) {
          constexpr int j = c + 2 + decltype(jc)::value;
          fmac_bcast<j>(d[j], nl, l);
          fmac_bcast<j>(p[j], nl, q);
        });
      }
    }
  });
}


// One wave per block.  LDS budget is exactly 160 KB / 8 = 20480 B per block so that 8 blocks (2 waves per SIMD) are
// resident per CU and one feature's dependent chains (the Cholesky) overlap another's:
//   phases A2/B:  rows of [J | C | E] (64 x 34 doubles, wave-uniform broadcast operands)          17408 B
//   phases C-E :  the factor L (packed lower triangle, 2112 doubles) in the SAME bytes            16896 B
//   all phases :  solved right-hand sides (64 x 4) + column exchange buffer (2 x 64)               3072 B
// B itself (written column by column in phase B, consumed block by block in phase C by the lane that wrote it) goes
// through a per-feature scratch in device memory: 16.9 KB per feature, 2.2 MB per XCD in flight = L2-resident.
//
// BORDERED (features with at most 30 observations, n + 4 <= 64): the four right-hand sides [r | H_f] ride along as four extra
// ROWS n..n+3 of the matrix being factorized.  The forward substitution then is the factorization's own row update (no
// per-column v_readlane / select code for right-hand sides), the sums y^T y, Z^T y, Z^T Z of the gate appear as the (negated)
// Schur complement in the 4x4 corner, and the left-looking update of a 16-column block against the finished columns is a
// product of LDS-resident panels - v_mfma_f64_16x16x4_f64 - instead of a 20-FMA-per-column scalar loop.  The variant
// without the border keeps the all-VALU path for 31 / 32 observations.
typedef double double4_t __attribute__((ext_vector_type(4)));

// Synchronisation inside one feature wave: LDS traffic of a wave is in order, so only the compiler (and the outstanding-
// counter waits of the fences) have to be told.  A real s_barrier would couple the eight independent feature waves of a
// fused workgroup.
#define OVP_WSYNC()                                          \
  do {                                                       \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   \
    __builtin_amdgcn_wave_barrier();                         \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");   \
  } while (0)

// phase stamps (s_memtime) for the "cycles" debug read: compiled in only with -DOVP_K1_STAMPS (they cost 20 VGPRs)
#ifdef OVP_K1_STAMPS
#define OVP_STAMP_DECL long long tstamp[8]
#define OVP_STAMP(i) tstamp[i] = __builtin_readcyclecounter()
#define OVP_STAMP_ONLY(...) __VA_ARGS__
#else
#define OVP_STAMP_DECL
#define OVP_STAMP(i)
#define OVP_STAMP_ONLY(...)
#endif

// ------------------------------------------------------------------------------------------------------------------------
// Round 5: the bordered factorization with B built block by block INSIDE the factorization - no per-feature scratch in device
// memory (Bscr: 17 KB per feature written and read back by the same wave, 34 MB per launch at 2000 features).
//
// What made the scratch necessary was the LDS budget (20 KB per feature wave, two waves per SIMD): the operand rows [J | C | E]
// of all 60 measurement rows (2040 doubles) and the packed factor (2112) do not fit side by side.  They do not have to:
//   * the operand rows of a 16-row tile are only read as the COLUMN side of block jb = tile (row side = registers), i.e. they are
//     dead once the block's 16 columns of B have been built;
//   * of the factor, a later block reads only the tiles BELOW the diagonal tile of a finished block column (left-looking update of
//     rows >= j0 with columns < j0); the diagonal tiles live and die in registers (k1_elim16), the corner is read from registers.
// So the six sub-diagonal 16 x 16 tiles of L move into the space the operand rows free, at fixed addresses (doubles):
//     rows of tile t   [544 t, 544 t + 544)      J 16x6 at +0, C 16x14 at +96, E 16x14 at +320   (tile 3: 12 rows packed as
//                                                 J 12x6 at +0, C at +72, E at +240 - it ends at 2040)
//     exchange tile    [2040, 2296)               256 doubles, XOR-swizzled (transposition of MFMA results, diagonal-tile
//                                                 broadcast, border staging; P_cc in phase A2)
//     L(1,0) 0   L(2,0) 256   L(3,0) 2296   |   L(2,1) 544   L(3,1) 800   |   L(3,2) 1088         (column-major 16 x 16)
//     corner           [1056, 1072)               the 4 x 4 Schur corner, column by column as the blocks finish
// L(rt,0) are written behind block 0's build (tile-0 rows dead; L(3,0) sits in the free tail), L(rt,1) behind block 1's, L(3,2)
// behind block 2's; the corner words lie in tile 1's rows, which are dead - or were never written - when the first corner
// column finishes (block n >> 4).  2552 of the 2560 doubles are spoken for at the worst moment.
static constexpr int V3_ROWT = 544;
static constexpr int V3_ST = 2040;
static constexpr int V3_CORNER = 1056;
__device__ __forceinline__ int v3_ltile(int rt, int ct) {
  return ct == 0 ? (rt == 1 ? 0 : (rt == 2 ? 256 : 2296)) : (ct == 1 ? (rt == 2 ? 544 : 800) : 1088);
}

// Phases B + C of a bordered feature (n = 2m <= 60 rows + 4 border rows): returns the gate's sums (sums[0] = y^T y, [1..3] = Z^T y,
// [4..9] = Z^T Z upper by rows) and spd.  jrow / crow / u: this lane's row of J, C and of E + C P_cc; the operand rows are in LDS.
__device__ __forceinline__ void bordered_factor_lds(const double* __restrict__ P, const int ldp, const int lane, const int n, const int m,
                                                    const bool valid, const int ida, const double (&jrow)[6], const double (&crow)[14],
                                                    const double (&u)[14], const double res, const double (&hf)[3],
                                                    double* const smem, bool& spd, double (&sums)[10], long long* const dbg2) {
  const int r = lane & 1;
  OVP_STAMP_ONLY(long long t_build = 0, t_elim = 0;)
  const int nb4 = n + 4;
  const int nblk = (nb4 + 15) >> 4;
  const int lr = lane >> 4, lc = lane & 15;
  double* const sT = smem + V3_ST;
  // the 6 x 6 block P[clone(b), clone(a)] is shared by the two lanes of observation a: lane r loads columns 3r..3r+2 of all six
  // rows, so that the PAIR reads one 48-byte row segment per request - half the cache lines per vector-memory instruction of a
  // split by rows (this phase is bound by the access rate of the vector L1, NOTES 5) - and only the lanes of the lower triangle
  // ask.  ONE buffer: the next observation's block is requested as soon as this one's has been folded into t, and arrives behind
  // the 68 FMAs of the LDS operands.
  double pc[18];
  auto fetch = [&](int b) {
    const int idb = __builtin_amdgcn_readlane(ida, 2 * b);  // lane 2b holds clone_id of observation b
    int coloff = ida + 3 * r;
    asm volatile("" : "+v"(coloff));  // recomputed per request: kept across the loop it is the one value the allocator spills
    const double* src = P + (idb * ldp + coloff);  // 32-bit element offsets: (idb + k) * ldp + ida + l < 2^31
    if (lane >= 2 * b && valid) {
#pragma unroll
      for (int k = 0; k < 6; ++k)
#pragma unroll
        for (int l = 0; l < 3; ++l) pc[3 * k + l] = src[k * ldp + l];
    }
  };
  // t = P[clone(b), clone(a)] j^T: each lane folds ITS three columns into both rows of the pair (its own j and the partner's,
  // fetched once) and the pair exchanges six partial sums - not eighteen covariance entries - per observation
  double jown[3], jpar[3];
#pragma unroll
  for (int l = 0; l < 3; ++l) {
    const double lo = swap_pair_f64(jrow[l]), hi = swap_pair_f64(jrow[3 + l]);
    jown[l] = r ? jrow[3 + l] : jrow[l];
    jpar[l] = r ? hi : lo;
  }
  fetch(0);
#pragma nounroll
  for (int jb = 0; jb < nblk; ++jb) {
    const int j0 = __builtin_amdgcn_readfirstlane(16 * jb);
    double ab[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) ab[t] = 0.0;  // upper triangle, corner columns, rows past the border
    // ---- columns j0 .. j0+15 of B = H_x P H_x^T + I, this lane's row (observations 8 jb .. 8 jb + 7)
    OVP_STAMP_ONLY(const long long tq0 = __builtin_readcyclecounter();)
    const double* const rows = smem + V3_ROWT * jb;
    const int offC = jb == 3 ? 72 : 96, offE = jb == 3 ? 240 : 320;  // (tile 3 holds 12 rows)
    static_for<8>([&](auto oc) {
      constexpr int o = decltype(oc)::value;
      const int b = 8 * jb + o;
      if (b < m) {  // (wave-uniform)
        double t[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          double ta = 0.0, tp = 0.0;
#pragma unroll
          for (int l = 0; l < 3; ++l) {
            ta = fma(jown[l], pc[3 * k + l], ta);
            tp = fma(jpar[l], pc[3 * k + l], tp);
          }
          t[k] = ta + swap_pair_f64(tp);
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) asm volatile("" : "+v"(t[k]));  // t complete before the buffer is refilled
        if (b + 1 < m) fetch(b + 1);
        __builtin_amdgcn_sched_barrier(0);
        const bool own = valid && lane >= 2 * b;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
          const int col = 2 * b + rr;
          // the operand row in two portions (C, then J and E): all 17 broadcast reads in flight at once cost 68 registers, and this
          // loop lives next to ab[16] and the covariance buffer; the SIMD's other wave covers the second round trip
          double s0 = 0.0, s1 = 0.0;
          {
            const double2_t* pcv = reinterpret_cast<const double2_t*>(rows + offC + 14 * (2 * o + rr));
            double2_t vc[7];
#pragma unroll
            for (int q = 0; q < 7; ++q) vc[q] = pcv[q];
#pragma unroll
            for (int k = 0; k < 14; ++k) s1 = fma(u[k], vc[k >> 1][k & 1], s1);
          }
          asm volatile("" : "+v"(s1));
          __builtin_amdgcn_sched_barrier(0);
          {
            const double2_t* pj = reinterpret_cast<const double2_t*>(rows + 6 * (2 * o + rr));
            const double2_t* pev = reinterpret_cast<const double2_t*>(rows + offE + 14 * (2 * o + rr));
            double2_t vj[3], ve[7];
#pragma unroll
            for (int q = 0; q < 3; ++q) vj[q] = pj[q];
#pragma unroll
            for (int q = 0; q < 7; ++q) ve[q] = pev[q];
#pragma unroll
            for (int k = 0; k < 6; ++k) s0 = fma(t[k], vj[k >> 1][k & 1], s0);
#pragma unroll
            for (int k = 0; k < 14; ++k) s0 = fma(crow[k], ve[k >> 1][k & 1], s0);
          }
          double bv = (s0 + s1) + (col == lane ? 1.0 : 0.0);
          asm volatile("" : "+v"(bv));  // finish this column before the next one's broadcast reads are issued (registers)
          __builtin_amdgcn_sched_barrier(0);
          ab[2 * o + rr] = own ? bv : 0.0;
        }
      }
    });
    OVP_STAMP_ONLY(t_build += __builtin_readcyclecounter() - tq0;)
    // ---- border rows n..n+3 of the block's columns: [r | H_f] of measurement row `col`, handed over through the exchange tile
    if (j0 < n) {
      if (valid && lr == jb) {
        double2_t* d = reinterpret_cast<double2_t*>(sT + 4 * lc);
        d[0] = double2_t{res, hf[0]};
        d[1] = double2_t{hf[1], hf[2]};
      }
      OVP_WSYNC();
      if (lane >= n && lane < nb4) {
        const double* s = sT + (lane - n);
        static_for<16>([&](auto tc) {
          constexpr int t = decltype(tc)::value;
          if (j0 + t < n) ab[t] = s[4 * t];
        });
      }
      OVP_WSYNC();
    }
    // ---- left-looking update of rows >= j0 with the finished columns 0..j0-1 (f64 MFMA over the sub-diagonal tiles)
    if (jb > 0) {
      double4_t acc[4];
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) acc[rt] = double4_t{0.0, 0.0, 0.0, 0.0};
      const int kend = j0 < n ? j0 : n;  // only factor columns (< n) contribute, never the corner columns
#pragma nounroll
      for (int kk = 0; kk < kend; kk += 4) {
        const int cb = kk >> 4;
        const int off = (((kk & 15) + lr) << 4) + lc;  // column (kk & 15) + lr of the tile, row lc
        const bool on = kk + lr < n;
        const double bv = on ? smem[v3_ltile(jb, cb) + off] : 0.0;
        double av[4];
#pragma unroll
        for (int rt = 1; rt < 4; ++rt) av[rt] = (rt > jb && rt < nblk && on) ? smem[v3_ltile(rt, cb) + off] : 0.0;
#pragma unroll
        for (int rt = 1; rt < 4; ++rt) {
          if (rt == jb) acc[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(bv, bv, acc[rt], 0, 0, 0);
          else if (rt > jb && rt < nblk) acc[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[rt], bv, acc[rt], 0, 0, 0);
        }
      }
      // C layout (row lr + 4 v, column lc) -> row-per-lane through the exchange tile, one row tile at a time; element (i, j) of the
      // tile sits at 16 i + (j ^ i): conflict-free both ways
      // (the swizzled addresses are formed where they are used, from laundered lane coordinates: as loop invariants they would
      // occupy 28 registers across the whole block loop)
#pragma unroll
      for (int rt = 1; rt < 4; ++rt) {
        if (rt >= jb && rt < nblk) {
          int lcv = lc, lrv = lr;
          asm volatile("" : "+v"(lcv), "+v"(lrv));
#pragma unroll
          for (int v = 0; v < 4; ++v) sT[((lrv + 4 * v) << 4) + (lcv ^ (lrv + 4 * v))] = acc[rt][v];
          OVP_WSYNC();
          if (lr == rt) {
            const int x = (lcv << 4) | lcv;  // (lc << 4) + (t ^ lc) == x ^ t
            static_for<16>([&](auto tc) {
              constexpr int t = decltype(tc)::value;
              ab[t] -= sT[x ^ t];
            });
          }
          OVP_WSYNC();
        }
      }
    }
    // ---- factor the block's columns < n (the border rows take part like any other row, the corner columns are never pivots):
    // the diagonal tile's rows go to every DPP row through the exchange tile, then one fused elimination
    const int ncol = n - j0 < 16 ? (n - j0 > 0 ? n - j0 : 0) : 16;  // (wave-uniform)
    OVP_STAMP_ONLY(const long long tq1 = __builtin_readcyclecounter();)
    if (ncol > 0) {
      int lcv = lc;
      asm volatile("" : "+v"(lcv));
      const int xq = (lcv << 3) | (lcv & 7);  // 16-byte unit (lc << 3) + (q ^ (lc & 7)) == xq ^ q
      if (lr == jb) {
        double2_t* dst = reinterpret_cast<double2_t*>(sT);
#pragma unroll
        for (int q = 0; q < 8; ++q) dst[xq ^ q] = double2_t{ab[2 * q], ab[2 * q + 1]};
      }
      OVP_WSYNC();
      double dd[16];
      {
        const double2_t* src = reinterpret_cast<const double2_t*>(sT);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const double2_t v = src[xq ^ q];
          dd[2 * q] = v[0];
          dd[2 * q + 1] = v[1];
        }
      }
      k1_elim16(dd, ab, ncol, spd);
      OVP_WSYNC();
    }
    OVP_STAMP_ONLY(t_elim += __builtin_readcyclecounter() - tq1;)
    // ---- publish: the block's tiles below the diagonal one (later blocks read nothing else), the corner columns' border rows
    if (jb + 1 < nblk && lr > jb) {
      double* const dst = smem + v3_ltile(lr, jb) + lc;
      static_for<16>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        dst[16 * t] = ab[t];
      });
    }
    if (j0 + 16 > n) {
      static_for<16>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        const int q = j0 + t - n;  // (wave-uniform) corner column
        if (q >= 0 && q < 4 && lane >= n && lane < nb4) smem[V3_CORNER + 4 * q + (lane - n)] = ab[t];
      });
    }
    OVP_WSYNC();
  }
  {
    auto corner = [&](int q, int qq) { return -smem[V3_CORNER + 4 * qq + q]; };  // q >= qq
    sums[0] = corner(0, 0);
    sums[1] = corner(1, 0);
    sums[2] = corner(2, 0);
    sums[3] = corner(3, 0);
    sums[4] = corner(1, 1);
    sums[5] = corner(2, 1);
    sums[6] = corner(3, 1);
    sums[7] = corner(2, 2);
    sums[8] = corner(3, 2);
    sums[9] = corner(3, 3);
  }
  OVP_STAMP_ONLY(if (dbg2 && lane == 0) {
    dbg2[0] = t_build;
    dbg2[1] = t_elim;
  })
}


// The work of one wave on feature f; smem = this wave's 20480 bytes of LDS.
// LDSB (bordered only): B is built block by block inside the factorization (bordered_factor_lds) - nothing goes through Bscr.
template <bool BORDERED, bool LDSB = false>
__device__ __forceinline__ void feat_body(const FeatParams& p, const int f, const int lane, double* const smem) {
  static_assert(BORDERED || !LDSB, "the LDS-resident build exists for the bordered factorization only");
  const int m = p.n_meas[f];
  const int n = 2 * m;
  const int so = p.slot ? p.slot[f] : f;            // where this feature's rows go in rec / G (-1: nowhere, it is not part of the update)
  const int nout = p.slot ? p.n_out : p.n_feats;

  double* const sJ = smem;                 // [64][6]
  double* const sC = smem + NR * 6;        // [64][14]
  double* const sE = sC + NR * 14;         // [64][14]   (ends at 2176)
  double* const sL = smem;                 // [LCOLS] factor, valid from phase C on (aliases sJ/sC/sE)
  double* const sY = smem + (LDSB ? V3_ST : 2176);  // [64][4]   (LDSB: only P_cc of phase A2 lives here)
  double* const sB = sY + NR * 4;          // [2][64]
  static_assert(LCOLS <= 2176, "factor must fit in the [J|C|E] region");
  double* const Bg = p.Bscr + (size_t)f * LCOLS;  // B = H_x P H_x^T + I, packed like sL

  const int a = lane >> 1, r = lane & 1;
  OVP_STAMP_DECL;
  OVP_STAMP(0);
  const bool valid = lane < n;
  // UpdaterMSCKF.cpp:94-96; features used by an accepted plane left feature_vec before the point loop (:657-666)
  const bool feat_ok = (m >= 2) && (m <= OVP_MAX_MEAS_DEV) && !(p.skip && p.skip[f]) && f >= p.range_lo && f < p.range_hi;
  const int* cidx = p.clone_idx + (size_t)f * p.max_meas;
  const int ci = cidx[valid ? a : 0];
  const int ida = p.clone_id[ci];

  // ------------------------------------------------------------------------------------------
  // Phase A: measurement model for observation a, keep row r.   UpdaterHelper.cpp:345-444
  // ------------------------------------------------------------------------------------------
  double jrow[6], crow[14], hf[3], res;
  build_bearing_row(p, f, a, r, valid, ci, jrow, crow, hf, res);
  OVP_STAMP(1);
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
      double* sPcc = sY;  // 196 doubles in the sY|sB region, which is not in use yet
      for (int idx = lane; idx < 196; idx += 64) {
        const int kk = idx / 14, k = idx - 14 * kk;
        const bool on = ((p.calmask >> kk) & 1) && ((p.calmask >> k) & 1);
        sPcc[idx] = on ? P[(size_t)p.calcol[kk] * ldp + p.calcol[k]] : 0.0;
      }
      // e = j P[clone(a), cal]: the two lanes of an observation need the same 14 x 6 block of P; each loads half
      // (7 calibration rows) and the halves are exchanged with a DPP pair swap
      double e[14];
      {
        double mine[42];
#pragma unroll
        for (int kk = 0; kk < 7; ++kk) {
          const int crow_id = r ? p.calcol[kk + 7] : p.calcol[kk];
          const double* prow = P + (size_t)crow_id * ldp + ida;
#pragma unroll
          for (int l = 0; l < 6; ++l) mine[6 * kk + l] = prow[l];
        }
#pragma unroll
        for (int kk = 0; kk < 7; ++kk) {
          double ea = 0.0, eb = 0.0;
#pragma unroll
          for (int l = 0; l < 6; ++l) {
            const double other = swap_pair_f64(mine[6 * kk + l]);
            ea = fma(jrow[l], mine[6 * kk + l], ea);
            eb = fma(jrow[l], other, eb);
          }
          e[kk] = r ? eb : ea;
          e[kk + 7] = r ? ea : eb;
        }
      }
      OVP_WSYNC();
      // u = e + c P_cc, one row of P_cc (7 x ds_read_b128 broadcasts) at a time; the scheduling barrier keeps the
      // compiler from hoisting all 98 reads (392 VGPRs) in front of the FMAs
#pragma unroll
      for (int k = 0; k < 14; ++k) u[k] = e[k];
      static_for<14>([&](auto kc) {
        constexpr int kk = decltype(kc)::value;
        const double2_t* prow = reinterpret_cast<const double2_t*>(sPcc + kk * 14);
        double2_t pv[7];
#pragma unroll
        for (int q = 0; q < 7; ++q) pv[q] = prow[q];
#pragma unroll
        for (int k = 0; k < 14; ++k) u[k] = fma(crow[kk], pv[k >> 1][k & 1], u[k]);
        __builtin_amdgcn_sched_barrier(0);
      });
      // pin u here: otherwise the FMAs are sunk behind the barrier into phase B and all of P_cc stays live in registers
#pragma unroll
      for (int k = 0; k < 14; ++k) asm volatile("" : "+v"(u[k]));
      if constexpr (LDSB) {
        // operand rows by 16-row tile (a tile dies with its block, see bordered_factor_lds); rows >= n have no storage
        if (valid) {
          double* const rt = smem + V3_ROWT * (lane >> 4);
          const int rl = lane & 15;
          const int offC = (lane >> 4) == 3 ? 72 : 96, offE = (lane >> 4) == 3 ? 240 : 320;  // (tile 3 holds 12 rows)
#pragma unroll
          for (int l = 0; l < 6; ++l) rt[6 * rl + l] = jrow[l];
#pragma unroll
          for (int k = 0; k < 14; ++k) {
            rt[offC + 14 * rl + k] = crow[k];
            rt[offE + 14 * rl + k] = e[k];
          }
        }
      } else {
#pragma unroll
        for (int l = 0; l < 6; ++l) sJ[lane * 6 + l] = jrow[l];
#pragma unroll
        for (int k = 0; k < 14; ++k) {
          sC[lane * 14 + k] = crow[k];
          sE[lane * 14 + k] = e[k];
        }
      }
    }
    OVP_WSYNC();
    OVP_STAMP(2);

    // ----------------------------------------------------------------------------------------
    // Phase B: row `lane` of B = H_x P H_x^T + I, lower triangle, packed in LDS
    // ----------------------------------------------------------------------------------------
    if constexpr (!LDSB) {
      // the 6 x 6 block P[clone(b), clone(a)] is shared by the two lanes of observation a: lane r loads rows 3r..3r+2.
      // The block of observation b+1 is fetched while b is processed; the loop is unrolled by two over a pair of
      // buffers so that the prefetch needs no register copies.
      const int rowoff = 3 * r * ldp + ida;  // 32-bit element offsets: (idb + 3r + k) * ldp + ida + l < 2^31
      auto fetch = [&](double (&dst)[18], int b) {
        const int idb = __builtin_amdgcn_readlane(ida, 2 * b);  // lane 2b holds clone_id of observation b
        const double* src = P + (idb * ldp + rowoff);
        // only the rows of the lower triangle (lane >= 2b) use the block: phase B is bound by the 64 B/clk of the vector
        // memory path (9 KB of 6x6 blocks per wave and observation), inactive lanes cost nothing there
        if (lane >= 2 * b && valid) {
#pragma unroll
          for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int l = 0; l < 6; ++l) dst[6 * k + l] = src[k * ldp + l];
        }
      };
      auto column_pair = [&](const double (&pc)[18], double (&pn)[18], int b) {
        if (b + 1 < m) fetch(pn, b + 1);
        double t[6];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          double ta = 0.0, tb = 0.0;
#pragma unroll
          for (int l = 0; l < 6; ++l) {
            const double other = swap_pair_f64(pc[6 * k + l]);
            ta = fma(jrow[l], pc[6 * k + l], ta);
            tb = fma(jrow[l], other, tb);
          }
          t[k] = r ? tb : ta;
          t[k + 3] = r ? ta : tb;
        }
        // rows 2b, 2b+1 of [J | C | E] (wave-uniform addresses: LDS broadcasts), fetched as 16-byte vectors, one row at a
        // time: with two waves per SIMD the other wave covers the LDS latency and the registers stay under 256
        double2_t bpair;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
          const int col = 2 * b + rr;
          double2_t vj[3], vc[7], ve[7];
          {
            const double2_t* pj = reinterpret_cast<const double2_t*>(sJ + 12 * b + 6 * rr);
            const double2_t* pcv = reinterpret_cast<const double2_t*>(sC + 28 * b + 14 * rr);
            const double2_t* pev = reinterpret_cast<const double2_t*>(sE + 28 * b + 14 * rr);
#pragma unroll
            for (int q = 0; q < 3; ++q) vj[q] = pj[q];
#pragma unroll
            for (int q = 0; q < 7; ++q) vc[q] = pcv[q];
#pragma unroll
            for (int q = 0; q < 7; ++q) ve[q] = pev[q];
          }
          double s0 = 0.0, s1 = 0.0;
#pragma unroll
          for (int k = 0; k < 6; ++k) s0 = fma(t[k], vj[k >> 1][k & 1], s0);
#pragma unroll
          for (int k = 0; k < 14; ++k) s1 = fma(u[k], vc[k >> 1][k & 1], s1);
#pragma unroll
          for (int k = 0; k < 14; ++k) s0 = fma(crow[k], ve[k >> 1][k & 1], s0);
          double bv = (s0 + s1) + (col == lane ? 1.0 : 0.0);
          asm volatile("" : "+v"(bv));  // finish this column before the next one's 17 broadcast reads are issued (registers)
          __builtin_amdgcn_sched_barrier(0);
          bpair[rr] = bv;
        }
        // both columns of the observation leave in ONE 16-byte store (scratch layout: column pairs interleaved, boff):
        // a vector-memory instruction costs the CU's one memory pipe ~30 cycles whatever its width
        if (lane >= 2 * b && (!BORDERED || lane < n)) *reinterpret_cast<double2_t*>(Bg + boff(b) + 2 * (lane - 2 * b)) = bpair;
      };
      double bufA[18], bufB[18];
      fetch(bufA, 0);
      for (int b = 0; b < m; b += 2) {
        column_pair(bufA, bufB, b);
        if (b + 1 < m) column_pair(bufB, bufA, b + 1);
      }
      if constexpr (BORDERED) {
        // border rows n..n+3 of column `lane`: r, H_f(:,0..2) of measurement row `lane`
        if (valid) {
          double* cj = Bg + boff(lane >> 1) + 2 * (n - (lane & ~1)) + (lane & 1);  // element (n + q, lane) at cj[2 q]
          cj[0] = res;
          cj[2] = hf[0];
          cj[4] = hf[1];
          cj[6] = hf[2];
        }
      }
    }
    OVP_WSYNC();
    OVP_STAMP(3);

    // ----------------------------------------------------------------------------------------
    // Phase C: Cholesky of B with the row in registers, fused forward substitution of [r | H_f]
    // ----------------------------------------------------------------------------------------
    // Left-looking, blocked by 16 columns: the block's 16 entries of this lane's row live in registers, the finished
    // columns of L are read back from LDS (own element + 16-wide broadcast), so the code is a compact rolled loop
    // (an earlier fully unrolled 64-step version was instruction-fetch bound).
    bool spd = true;
    double sums[10];
    if constexpr (LDSB) {
      bordered_factor_lds(P, ldp, lane, n, m, valid, ida, jrow, crow, u, res, hf, smem, spd, sums,
                          p.dbg_cycles ? p.dbg_cycles + (size_t)p.n_feats * 8 + 2 * f : nullptr);
      OVP_STAMP(3);
      OVP_STAMP(4);
    } else if constexpr (BORDERED) {
      const int nb4 = n + 4;                      // rows of the bordered matrix
      const bool brow = lane < nb4;               // this lane holds one of them
      const int nblk = (nb4 + 15) >> 4;
      const int lr = lane >> 4, lc = lane & 15;   // MFMA operand coordinates
      double* const sT = sY;                       // 16 x 17 transpose scratch (sY | sB region, 3 KB)
#pragma nounroll
      for (int jb = 0; jb < nblk; ++jb) {
        const int j0 = __builtin_amdgcn_readfirstlane(16 * jb);
        double ab[16];
        static_for<8>([&](auto tc) {
          constexpr int t = 2 * decltype(tc)::value;
          const int col = j0 + t;  // even: the column pair (col, col + 1) is one 16-byte load
          double2_t v = {0.0, 0.0};  // upper triangle, the 4x4 corner and the padding start at zero
          if (brow && col < n && lane >= col) v = *reinterpret_cast<const double2_t*>(Bg + boff(col >> 1) + 2 * (lane - col));
          ab[t] = v[0];
          ab[t + 1] = v[1];
        });
        if (jb > 0) {
          // left-looking update of rows >= j0, columns j0..j0+15 with the finished columns 0..j0-1:
          //   D[rt] = L[16 rt .. 16 rt + 15][0 .. j0) * L[j0 .. j0 + 15][0 .. j0)^T     (one 16 x 16 tile per row tile rt >= jb)
          // A operand: lane -> (row lc of the tile, column kk + lr); B operand: (row j0 + lc, column kk + lr).
          double4_t acc[4];
#pragma unroll
          for (int rt = 0; rt < 4; ++rt) acc[rt] = double4_t{0.0, 0.0, 0.0, 0.0};
          int col = lr;                              // column kk + lr of this lane
          int ck = coff(col) - (col & ~1);           // its offset; four columns further: + 248 - 4 col
          const int kend = j0 < n ? j0 : n;          // only factor columns (< n) contribute, never the corner columns
#pragma nounroll
          for (int kk = 0; kk < kend; kk += 4) {
            const double bv = (col < n) ? sL[ck + j0 + lc] : 0.0;
            double av[4];
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) av[rt] = (rt >= jb) ? sL[ck + 16 * rt + lc] : 0.0;
#pragma unroll
            for (int rt = 0; rt < 4; ++rt)
              if (rt >= jb) acc[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[rt], bv, acc[rt], 0, 0, 0);
            ck += 248 - 4 * col;
            col += 4;
          }
          // C layout (row lr + 4 v, column lc) -> row-per-lane through a 16 x 17 LDS tile, one row tile at a time
#pragma unroll
          for (int rt = 0; rt < 4; ++rt) {
            if (rt >= jb) {
#pragma unroll
              for (int v = 0; v < 4; ++v) sT[(lr + 4 * v) * 17 + lc] = acc[rt][v];
              __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
              __builtin_amdgcn_wave_barrier();
              __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
              if ((lane >> 4) == rt) {
                static_for<16>([&](auto tc) {
                  constexpr int t = decltype(tc)::value;
                  ab[t] -= sT[lc * 17 + t];
                });
              }
              __builtin_amdgcn_wave_barrier();
            }
          }
        }
        // factor the block's columns < n: the border rows take part like any other row, the corner columns are never pivots.
        // The diagonal tile's rows go to every DPP row through the transpose scratch, then one fused elimination (k1_elim16).
        const int ncol = n - j0 < 16 ? (n - j0 > 0 ? n - j0 : 0) : 16;  // (wave-uniform)
        if (ncol > 0) {
          __builtin_amdgcn_wave_barrier();
          if ((lane >> 4) == jb) {
            double2_t* dst = reinterpret_cast<double2_t*>(sT + lc * 18);
#pragma unroll
            for (int q = 0; q < 8; ++q) dst[q] = double2_t{ab[2 * q], ab[2 * q + 1]};
          }
          OVP_WSYNC();
          double dd[16];
          {
            const double2_t* src = reinterpret_cast<const double2_t*>(sT + lc * 18);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
              const double2_t v = src[q];
              dd[2 * q] = v[0];
              dd[2 * q + 1] = v[1];
            }
          }
          k1_elim16(dd, ab, ncol, spd);
          // rows above the diagonal do not belong to a column: the diagonal tile's own rows end with garbage there
          if ((lane >> 4) == jb) {
            static_for<16>([&](auto tc) {
              constexpr int t = decltype(tc)::value;
              if (t > lc && t < ncol) ab[t] = 0.0;
            });
          }
          __builtin_amdgcn_wave_barrier();
        }
        // publish the block (factor columns, and for the last block the corner)
        static_for<16>([&](auto tc) {
          constexpr int t = decltype(tc)::value;
          const int col = j0 + t;
          if (lane >= (col & ~1)) sL[coff(col) + lane - (col & ~1)] = ab[t];
        });
        OVP_WSYNC();
      }
      // corner (rows / columns n..n+3) = -[y Z]^T [y Z]: sums[0] = y^T y, [1..3] = Z^T y, [4..9] = Z^T Z (upper by rows)
      {
        auto corner = [&](int q, int qq) {  // q >= qq
          const int col = n + qq;
          return -sL[coff(col) + (n + q) - (col & ~1)];
        };
        sums[0] = corner(0, 0);
        sums[1] = corner(1, 0);
        sums[2] = corner(2, 0);
        sums[3] = corner(3, 0);
        sums[4] = corner(1, 1);
        sums[5] = corner(2, 1);
        sums[6] = corner(3, 1);
        sums[7] = corner(2, 2);
        sums[8] = corner(3, 2);
        sums[9] = corner(3, 3);
      }
      OVP_STAMP(4);
    } else {
      double rh0 = res, rh1 = hf[0], rh2 = hf[1], rh3 = hf[2];
      const int nblk = (n + 15) >> 4;
      OVP_STAMP_ONLY(long long t_ll = 0, t_ib = 0;)
#pragma nounroll
      for (int jb = 0; jb < nblk; ++jb) {
        const int j0 = __builtin_amdgcn_readfirstlane(16 * jb);  // scalar: lane selects below must not waterfall
        double ab[16];
        static_for<16>([&](auto tc) {
          constexpr int t = decltype(tc)::value;
          const int col = j0 + t;
          double v = (col == lane) ? 1.0 : 0.0;                       // identity padding for rows/columns >= n
          if (valid && col < n && lane >= (col & ~1)) v = Bg[boff(col >> 1) + 2 * (lane - (col & ~1)) + (col & 1)];
          ab[t] = v;
        });
        OVP_STAMP_ONLY(long long tq0 = __builtin_readcyclecounter();)
        // update with the finished columns k < j0
        int ck = 0;  // coff(k) - (k & ~1), advanced incrementally: +64-k after an even column, +63-k after an odd one
#pragma nounroll
        for (int k = 0; k < j0; ck += 64 - k - (k & 1), ++k) {
          double lik = sL[ck + lane];
          if (lane < j0) lik = 0.0;  // rows of finished blocks are final (k < j0 <= lane also covers the upper triangle)
          // L[j0 .. j0+15][k] and y_k: wave-uniform, 16-byte aligned (column starts are even) -> ds_read_b128 broadcasts
          const double2_t* lj = reinterpret_cast<const double2_t*>(sL + ck + j0);
          const double2_t* yk = reinterpret_cast<const double2_t*>(sY + 4 * k);
          double2_t lv[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) lv[q] = lj[q];
          const double2_t y01 = yk[0], y23 = yk[1];
          static_for<16>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            ab[t] = fma(-lik, lv[t >> 1][t & 1], ab[t]);
          });
          const double lir = (lane < j0 + 16) ? lik : 0.0;  // right-hand sides: only the rows of this block
          rh0 = fma(-lir, y01[0], rh0);
          rh1 = fma(-lir, y01[1], rh1);
          rh2 = fma(-lir, y23[0], rh2);
          rh3 = fma(-lir, y23[1], rh3);
        }
        OVP_STAMP_ONLY(long long tq1 = __builtin_readcyclecounter(); t_ll += tq1 - tq0;)
        // factor the block: 16 right-looking steps inside the registers.  Column c of L is broadcast to the other rows
        // through a small LDS buffer (one ds_write + a few 16-byte broadcast reads per step instead of 2 v_readlane per
        // element); the next pivot is taken with a look-ahead so its rsq/Newton chain overlaps the LDS round trip.
        double piv = readlane_f64(ab[0], j0);
        double inv = rsqrt_nr(piv);
        static_for<16>([&](auto cc) {
          constexpr int c = decltype(cc)::value;
          const int kg = j0 + c;  // global column, wave-uniform
          spd = spd && (piv > 0.0);
          const double inv_c = inv;
          double l = ab[c] * inv_c;
          if (lane < kg) l = 0.0;  // rows above the diagonal do not belong to column kg
          ab[c] = l;
          double* cb = sB + (c & 1) * NR;
          cb[lane] = l;
          if constexpr (c + 1 < 16) {
            const double l1 = readlane_f64(l, kg + 1);
            ab[c + 1] = fma(-l, l1, ab[c + 1]);
            piv = readlane_f64(ab[c + 1], kg + 1);
            inv = rsqrt_nr(piv);
          }
          const double x0 = readlane_f64(rh0, kg) * inv_c, x1 = readlane_f64(rh1, kg) * inv_c;
          const double x2 = readlane_f64(rh2, kg) * inv_c, x3 = readlane_f64(rh3, kg) * inv_c;
          if (lane > kg && lane < j0 + 16) {  // rows of later blocks receive this column through the left-looking pass
            rh0 = fma(-l, x0, rh0);
            rh1 = fma(-l, x1, rh1);
            rh2 = fma(-l, x2, rh2);
            rh3 = fma(-l, x3, rh3);
          } else if (lane == kg) {
            rh0 = x0;
            rh1 = x1;
            rh2 = x2;
            rh3 = x3;
          }
          if constexpr (c + 2 < 16) {
            // l_(j0+j), j = c+2..15, read as aligned pairs starting at the even index <= c+2
            constexpr int e0 = (c + 2) & ~1;
            const double2_t* cbv = reinterpret_cast<const double2_t*>(cb + j0 + e0);
            double2_t lv[(16 - e0) / 2];
#pragma unroll
            for (int q = 0; q < (16 - e0) / 2; ++q) lv[q] = cbv[q];
            static_for<14 - c>([&](auto jc) {
              constexpr int j = c + 2 + decltype(jc)::value;
              ab[j] = fma(-l, lv[(j - e0) >> 1][(j - e0) & 1], ab[j]);
            });
          }
        });
        OVP_STAMP_ONLY(t_ib += __builtin_readcyclecounter() - tq1;)
        // publish the block's columns of L and the solved right-hand sides of its rows
        static_for<16>([&](auto tc) {
          constexpr int t = decltype(tc)::value;
          const int col = j0 + t;
          if (lane >= (col & ~1)) sL[coff(col) + lane - (col & ~1)] = ab[t];
        });
        if (lane >= j0 && lane < j0 + 16) {
          sY[4 * lane + 0] = rh0;
          sY[4 * lane + 1] = rh1;
          sY[4 * lane + 2] = rh2;
          sY[4 * lane + 3] = rh3;
        }
        OVP_WSYNC();
      }
      OVP_STAMP(4);
      OVP_STAMP_ONLY(if (p.dbg_cycles && lane == 0) {
        p.dbg_cycles[(size_t)p.n_feats * 8 + 2 * f] = t_ll;
        p.dbg_cycles[(size_t)p.n_feats * 8 + 2 * f + 1] = t_ib;
      })
      yv = valid ? rh0 : 0.0;
      zv[0] = valid ? rh1 : 0.0;
      zv[1] = valid ? rh2 : 0.0;
      zv[2] = valid ? rh3 : 0.0;

      // ----------------------------------------------------------------------------------------
      // Phase D: chi2 = y^T y - (Z^T y)^T (Z^T Z)^-1 (Z^T y)      UpdaterMSCKF.cpp:739-764
      // ----------------------------------------------------------------------------------------
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
  OVP_WSYNC();
  OVP_STAMP(5);

  // ------------------------------------------------------------------------------------------
  // Phase E: projector rows G = Q1^T H_x, g = Q1^T r  (Q1 by CholeskyQR2 on H_f), staged in LDS
  // ------------------------------------------------------------------------------------------
  const int ldg = p.ldg;
  double* Gst = sL;  // 3 * ldg doubles (ldg <= OVP_LDG_CAP = LCOLS / 3); the factor is no longer needed
  for (int idx = lane; idx < 3 * ldg; idx += 64) Gst[idx] = 0.0;
  OVP_WSYNC();
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
  OVP_WSYNC();
  OVP_STAMP(6);
  if (so >= 0) {
    double* gout = p.G + (size_t)3 * so * ldg;
    for (int idx = lane; idx < 3 * ldg; idx += 64) gout[idx] = Gst[idx];
  }

  // ------------------------------------------------------------------------------------------
  // sparse rows for K2 (zero for rejected features / unobserved clones)
  // ------------------------------------------------------------------------------------------
  unsigned long long seen = 0ull;
  if (accept) {
    seen = wave_or_u64(valid ? (1ull << ci) : 0ull);  // (a rolled loop over cidx[] was m dependent loads)
    if (valid) {
      double* ro = p.rec + (((size_t)ci * nout + so) * 2 + r) * OVP_REC;  // (accepted: so >= 0)
#pragma unroll
      for (int l = 0; l < 6; ++l) ro[l] = jrow[l];
#pragma unroll
      for (int k = 0; k < 14; ++k) ro[6 + k] = crow[k];
      ro[20] = res;
    }
  }
  if (so >= 0) {
    for (int cc = 0; cc < p.n_clones; ++cc) {
      if (!((seen >> cc) & 1ull)) {
        double* ro = p.rec + (((size_t)cc * nout + so) * 2) * OVP_REC;
        if (lane < 2 * OVP_REC) ro[lane] = 0.0;
      }
    }
  }
  OVP_STAMP(7);
  OVP_STAMP_ONLY(if (p.dbg_cycles && lane == 0) {
    for (int k = 0; k < 8; ++k) p.dbg_cycles[(size_t)f * 8 + k] = tstamp[k];
  })
  if (lane == 0) {
    p.chi2[f] = chi2;
    p.accept[f] = accept ? 1 : 0;
  }
}

template <bool BORDERED, bool LDSB>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_feat_gate(const FeatParams p) {
  __builtin_amdgcn_s_setprio(3);
  __shared__ __attribute__((aligned(16))) double smem[2560];
  feat_body<BORDERED, LDSB>(p, blockIdx.x, threadIdx.x, smem);
}

// Fused launch: workgroup 0 factorizes P (chol(P) does not depend on the measurements, but its eight latency-bound waves
// must not share SIMDs with feature waves: beside a one-wave-per-block K1 they stretch the slowest feature block - and the
// kernel - by 40 %; CU masks are kept symmetric per shader engine by the driver, so a side stream cannot be pinned to one
// CU either).  Every workgroup asks for all 160 KB of LDS, hence owns a CU: workgroup 0 has one to itself, the others run
// eight feature waves (two per SIMD, 20 KB of LDS each) exactly as the one-wave blocks did.  Feature f is handled by wave
// f / nwg of workgroup 1 + f % nwg, so small batches spread one wave per SIMD before they double up.
template <int MAXSLOT, bool LDSB>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_feat_chol(const FeatParams p,
                                                                                             const CholJob c) {
  __shared__ __attribute__((aligned(16))) double smem[8 * 2560];
  static_assert(tilechol_lds_doubles(OVP_TC_MAX_TILES) <= 8 * 2560, "chol(P) must fit in the workgroup's LDS");
  if (blockIdx.x == 0) {
    if (c.n > 0) tilechol_body<MAXSLOT>(c.A, c.L, c.Dinv, c.Lpack, c.n, c.ld, c.flag, 0, 0, smem, c.flip, c.boost, c.boost_n, c.boost_rel);  // (n = 0: a factor of P is at hand)
    return;
  }
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwg = gridDim.x - 1;
  const int f = wave * nwg + (blockIdx.x - 1);
  if (f >= p.n_feats) return;
  __builtin_amdgcn_s_setprio(3);
  feat_body<true, LDSB>(p, f, threadIdx.x & 63, smem + wave * 2560);
}

}  // namespace ovp

// OVP_K1_BSCR=1: the bordered factorization with B through the per-feature scratch in device memory (rounds 1-4), for A/B runs
static bool k1_through_scratch() { return getenv("OVP_K1_BSCR") != nullptr; }  // (read per call: the tests switch it)

extern "C" hipError_t ovp_launch_feat_gate(const ovp::FeatParams* p, hipStream_t stream) {
  if (p->n_feats <= 0) return hipSuccess;
  if (p->max_meas <= 30) {
    if (k1_through_scratch())
      hipLaunchKernelGGL((ovp::k_feat_gate<true, false>), dim3(p->n_feats), dim3(64), 0, stream, *p);
    else
      hipLaunchKernelGGL((ovp::k_feat_gate<true, true>), dim3(p->n_feats), dim3(64), 0, stream, *p);
  } else {
    hipLaunchKernelGGL((ovp::k_feat_gate<false, false>), dim3(p->n_feats), dim3(64), 0, stream, *p);
  }
  return hipGetLastError();
}

// 1 if ovp_launch_feat_chol can take this batch / matrix (otherwise: ovp_launch_feat_gate + ovp_launch_tilechol)
extern "C" int ovp_feat_chol_supported(const ovp::FeatParams* p, int n) {
  const int nt = (n + 15) >> 4;
  return p->n_feats > 0 && p->max_meas <= 30 && nt <= OVP_TC_MAX_TILES;
}

// features the fused-shape launch takes in ONE round while leaving a CU per XCD to a kernel on another stream
extern "C" int ovp_feat_chol_side_capacity(void) { return 247 * 8; }

extern "C" hipError_t ovp_launch_feat_chol(const ovp::FeatParams* p, const ovp::CholJob* c_in, hipStream_t stream) {
  const ovp::CholJob* c = c_in;
  const int F = p->n_feats;
  // feature workgroups of one round: 256 CUs, one of them factorizes.  Without a factorization in workgroup 0 (c_in->n == 0 while the
  // caller runs chol(P) as a kernel of its own on a side stream, see ovp_feat_chol_side_capacity) the round is 247 workgroups:
  // workgroups go to the eight XCDs in turn from an offset that changes between launches, so 248 of them leave every XCD a CU
  // for the other kernel's workgroup wherever it lands - 255 leave one CU on one XCD, and the two launches run one after the other
  const int cus = c_in->n == 0 && F <= ovp_feat_chol_side_capacity() ? 247 : 255;
  const int nwg = F <= cus ? F : (F <= 8 * cus ? cus : (F + 7) / 8);
  const int nt = (c_in->n + 15) >> 4;
  const int slots = (nt * (nt + 1) / 2 + ovp::TC_TILE_WAVES - 1) / ovp::TC_TILE_WAVES;
  if (k1_through_scratch()) {
    if (slots <= 15)
      hipLaunchKernelGGL((ovp::k_feat_chol<15, false>), dim3(nwg + 1), dim3(512), 0, stream, *p, *c);
    else if (slots <= 18)
      hipLaunchKernelGGL((ovp::k_feat_chol<18, false>), dim3(nwg + 1), dim3(512), 0, stream, *p, *c);
    else
      hipLaunchKernelGGL((ovp::k_feat_chol<25, false>), dim3(nwg + 1), dim3(512), 0, stream, *p, *c);
  } else {
    if (slots <= 15)
      hipLaunchKernelGGL((ovp::k_feat_chol<15, true>), dim3(nwg + 1), dim3(512), 0, stream, *p, *c);
    else if (slots <= 18)
      hipLaunchKernelGGL((ovp::k_feat_chol<18, true>), dim3(nwg + 1), dim3(512), 0, stream, *p, *c);
    else
      hipLaunchKernelGGL((ovp::k_feat_chol<25, true>), dim3(nwg + 1), dim3(512), 0, stream, *p, *c);
  }
  return hipGetLastError();
}
