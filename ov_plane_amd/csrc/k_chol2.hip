// Register-resident tile Cholesky, second generation: no factor wave, no inverse, no panel product.
//
// k_tilechol (k_tile.hip) spends its step in a serial chain: 16x16 Cholesky of the diagonal block by one wave, then its inverse,
// a workgroup barrier, the panel product with the inverse, another barrier, the trailing update (3.5 of 5.4 us per 16 columns
// at N = 240).  Here every wave that owns panel tiles of the current tile column runs the 16-column elimination ITSELF, on a
// private copy of the diagonal block AND on up to four of its panel tiles at once:
//   lane = 16 g + r : DPP row g works on panel tile g of this wave, r = row inside the tile; registers d[16] = row r of the
//   diagonal block (identical in the four DPP rows and in every wave), p[16] = row r of the panel tile;
//   column c:  l = d[c] / sqrt(pivot)  (pivot = lane c's d[c], one v_mov_b64_dpp row_newbcast),  q = p[c] / sqrt(pivot),
//              d[j] -= l * L_jc,  p[j] -= q * L_jc   with L_jc = lane j's l taken INSIDE the FMA (v_fmac_f64_dpp row_newbcast:j).
// The panel solve X L_kk^T = A_ik thus finishes with the last column of the diagonal block: no inverse, no second barrier, no
// cross-lane traffic besides the DPP operand.  Redundant factorizations of the diagonal block cost nothing (the waves would
// wait for it anyway) and are bit-identical.  The trailing update stays on v_mfma_f64_16x16x4_f64 with operands from LDS.
//
// Bordered right-hand side: an extra row `brow` below the matrix is factorized along (its tiles are ordinary panel tiles), so
// z = L^-1 brow^T and |z|^2 come for free; a right-looking... (see chol2_backsolve) gives L^-T z without leaving the registers.
// Modes (Chol2Job::mode): 0 = factor only (packed factor / dense factor / z written out), 1 = plane loop, update part
// (gate, back substitution, dx = L0 y, commit of the state tables), 2 = plane loop, range part (|Lr^-1 bn|^2, rank).
#include "k_tile_body.h"
#include "k_dpp.h"
#include "k_chol2.h"
#include <cstring>

namespace ovp {

static constexpr int C2_TS = 18;             // LDS row pitch of a 16x16 tile (doubles): rows 16-byte aligned
static constexpr int C2_TSZ = 16 * C2_TS;    // doubles per LDS tile
static constexpr int C2_WAVES = 12;          // 4 elimination waves + 8 tile waves
static constexpr int C2_EW = 4;
static constexpr int C2_TW = C2_WAVES - C2_EW;

typedef double dbl2_t __attribute__((ext_vector_type(2)));

// Fused elimination of a 16x16 diagonal block (d: row r = lane & 15, a copy in each DPP row) and of one panel tile per DPP row
// (p).  On return d = row r of L_kk (entries above the diagonal are garbage), p = row r of A_ik L_kk^-T, pv[c] = pivot of column c
// before the square root (lane-uniform).  Returns true if a pivot was not positive.
// pw: row r of the panel tile's LDS image (EVERY lane has one - lanes without a tile of their own mirror another DPP row's or
// point at scratch, see the caller); a pair of columns is stored as soon as it is final, so the publication of the panel rides in
// the gaps of the chain instead of following it.
// What the chain must not contain (tools/dpp64_bench.hip, cycles per 16 columns on one wave: 2460 for the bare elimination):
// a store under a lane mask per column (the pivot by lane 0: +240; the panel pairs by the lanes that have a tile: +800 over the
// unmasked stores - every masked block is an EXEC round trip through the scalar unit), a compare whose result goes through the
// scalar unit back into a select (the pivot floor test as `floor > 0 && !(piv >= floor)`: +1000 together with the per-column
// `bad`).  So: the pivots are stored pair by pair by EVERY lane (same words, same values - the waves' copies of the diagonal block
// are identical; WRITE_PIV - only where somebody reads them), only the last one is tested, the floor test exists only in the HAS_FLOOR instantiation and selects on VCC, the
// panel stores are unconditional.
template <bool HAS_FLOOR, bool WRITE_PIV, bool WRITE_D>
__device__ __forceinline__ bool fused_elim16(double (&d)[16], double (&p)[16], dbl2_t* __restrict__ pvw, const double floor,
                                             dbl2_t* __restrict__ pw, dbl2_t* __restrict__ dw, const int r) {
  double piv = bcast_row<0>(d[0]), piv_prev = 0.0;
  sfor<16>([&](auto cc) {
    constexpr int c = decltype(cc)::value;
    // floor: a pivot below it marks a direction the (positive semi-definite) matrix does not determine - the column is
    // dropped (no elimination with it, zero entry in the solution) instead of being divided by
    // 1 / sqrt(pivot): v_rsq_f64 (~2^-26 relative) and ONE Newton step (-> a couple of ulp; the second step of rsq_nr2 is
    // invisible next to the rounding of the updates that follow every pivot - tests: factor against numpy to 1e-13), folded into the
    // scaling: with y0 = rsq(piv), e = 0.5 - 0.5 piv y0^2 the scaled entries are d y0 (1 + e), and d y0, p y0, -d y0 are formed
    // beside e.  Dependent f64 operations from the pivot to the first update of the next one: rsq, piv y0 / 2, e, l - four instead
    // of eight (two Newton steps, then the scaling, then the negation): this chain is what a tile step consists of.
    double y0 = __builtin_amdgcn_rsq(piv);
    if constexpr (HAS_FLOOR) y0 = (piv >= floor) ? y0 : 0.0;
    const double hy = (0.5 * piv) * y0;
    const double ly = d[c] * y0, py = p[c] * y0;
    const double e = fma(-hy, y0, 0.5);
    const double l = fma(ly, e, ly);
    const double q = fma(py, e, py);
    // nl = -l is the DPP operand of every update of this column.  A DPP read of a VGPR needs two wait states behind the VALU
    // write; the hazard recognizer does not look inside asm statements, and non-volatile asm statements may be emitted in any
    // order - so the wait states go WITH THE WRITE (they used to sit in front of the first consumer in source order only: a
    // later consumer hoisted above it read the previous column's value - it happened, in one instantiation, after an unrelated
    // change of the code behind the chain)
    double nl;
    asm("v_fma_f64 %0, -%1, %2, -%1\n\ts_nop 1" : "=v"(nl) : "v"(ly), "v"(e));
    d[c] = l;
    p[c] = q;
    if constexpr (WRITE_PIV && (c & 1) == 1) pvw[c >> 1] = dbl2_t{piv_prev, piv};  // (every lane, the same words, the same values)
    piv_prev = piv;
    if constexpr (c + 1 < 16) {
      // next pivot first: its broadcast / rsq / Newton chain then overlaps the remaining updates of this column
      fmac_bcast<c + 1>(d[c + 1], nl, l);
      piv = bcast_row<c + 1>(d[c + 1]);
      fmac_bcast<c + 1>(p[c + 1], nl, q);
      sfor<14 - c>([&](auto jc) {
        constexpr int j = c + 2 + decltype(jc)::value;
        fmac_bcast<j>(d[j], nl, l);
        fmac_bcast<j>(p[j], nl, q);
      });
    }
    if constexpr ((c & 1) == 1) pw[c >> 1] = dbl2_t{p[c - 1], p[c]};
    // WRITE_D (one wave per step): row r of L_kk, zero above the diagonal, pair by pair like the panel (the four DPP rows write the
    // same words) - behind the chain these stores and their selects were 800 cycles in front of the wave's next step
    if constexpr (WRITE_D && (c & 1) == 1) dw[c >> 1] = dbl2_t{(c - 1 <= r) ? d[c - 1] : 0.0, (c <= r) ? d[c] : 0.0};
  });
  // a pivot that is not positive (or NaN) turns everything behind it into NaN - 1 / sqrt of it scales the whole column, and every
  // later update multiplies by an entry of that column - so the last pivot tells for all sixteen.  (With a floor dropped columns
  // stop the propagation, and the caller does not ask.)
  return HAS_FLOOR ? false : !(piv > 0.0);
}

// acc -= X Y^T for two row-major LDS tiles (pitch C2_TS).  The sum over the 16 inner indices is split over the four MFMAs as
// k = 4 lr + q (q = MFMA, lr = the lane's row group) instead of k = 4 q + lr: both operands use the same assignment, so the product
// is the same sum, and a lane's four operand values of a tile are 32 contiguous bytes - two 16-byte LDS reads per operand and tile
// instead of four 8-byte ones (the trailing update of the first tile columns is what the elimination waves wait for).
// `off` = the lane's place inside a tile image in BYTES, 8 (lc * C2_TS + 4 * lr) - formed once per step by the caller: as `lc, lr` arguments
// the compiler rebuilt it for every tile (a 16-cycle v_mul_lo_u32 among them), and VALU instructions of a tile wave queue behind the
// 64-cycle f64 MFMAs of the other tile wave on the SIMD.  The sign rides on the MFMA's operand modifier (for the f64 MFMAs the BLGP
// field is NEG[a, b, c]) instead of four v_xor per tile, and all four operand pieces are requested BEFORE the first MFMA: left to
// itself the compiler fetched the second half behind the first two MFMAs - two LDS round trips per tile instead of one.
__device__ __forceinline__ double4_t c2_mfma_xyT(const double* X, const double* Y, double4_t acc, int off) {
  const dbl2_t* xp = reinterpret_cast<const dbl2_t*>(reinterpret_cast<const char*>(X) + off);
  const dbl2_t* yp = reinterpret_cast<const dbl2_t*>(reinterpret_cast<const char*>(Y) + off);
  dbl2_t x0 = xp[0], y0 = yp[0], x1 = xp[1], y1 = yp[1];
  asm volatile("" : "+v"(x0), "+v"(y0), "+v"(x1), "+v"(y1));
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x0[0], y0[0], acc, 0, 0, 1);
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x0[1], y0[1], acc, 0, 0, 1);
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x1[0], y1[0], acc, 0, 0, 1);
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x1[1], y1[1], acc, 0, 0, 1);
  return acc;
}

// The last update of a column by the wave that eliminates it (chol2_factor, elimination waves): the wave's NH panel tiles
// (tile rows i0, i0 + 4, ...) and the diagonal block of column k get  -= P_i P_k^T  with panel k - 1 (buffer pbp), in place in their
// row-major LDS images (the diagonal block: Dbuf -> Dupd).  All operand reads first, then the four MFMAs of every tile round by
// round: NH + 1 independent accumulation chains, so an MFMA never waits for its predecessor (a dependent f64 MFMA starts ~200
// cycles behind it, an independent one ~110).  With `if (tile exists)` around each product the chains ran one after the other.
template <int NH>
__device__ __forceinline__ void c2_last_update(double* Dbuf, double* Dupd, double* Lim, const double* pbp, int k, int i0, int g, int r,
                                               bool skip) {
  const double* Pk = pbp + k * C2_TSZ;
  const dbl2_t* yp = reinterpret_cast<const dbl2_t*>(Pk + r * C2_TS + 4 * g);
  const dbl2_t y0 = yp[0], y1 = yp[1];
  double4_t acc[NH + 1];
  dbl2_t x0[NH + 1], x1[NH + 1];
#pragma unroll
  for (int t = 0; t < NH; ++t) {
    const double* im = Lim + (i0 + C2_EW * t) * C2_TSZ;
#pragma unroll
    for (int v = 0; v < 4; ++v) acc[t][v] = im[(g + 4 * v) * C2_TS + r];
    const dbl2_t* xp = reinterpret_cast<const dbl2_t*>(pbp + (i0 + C2_EW * t) * C2_TSZ + r * C2_TS + 4 * g);
    x0[t] = xp[0];
    x1[t] = xp[1];
  }
#pragma unroll
  for (int v = 0; v < 4; ++v) acc[NH][v] = Dbuf[(g + 4 * v) * C2_TS + r];
  x0[NH] = y0;
  x1[NH] = y1;
  if (!skip) {
    // The diagonal block (index NH) is computed by EVERY elimination wave and the copies must agree to the bit (they all go into
    // Dupd, and each wave reads its rows back from whatever landed last): it always takes two accumulation chains of two MFMAs,
    // whatever NH is.  A wave with at most one panel tile (the last steps, where the chain is all there is) splits that tile's
    // sum the same way, so that four independent MFMAs are in flight there as well; a tile belongs to one wave, so its
    // rounding is a function of the step only.
    constexpr int NS = NH <= 1 ? NH + 1 : 1;   // tiles with a second chain: the panel tile (if split) and the diagonal block
    constexpr int S0 = NH + 1 - NS;            // ... are the last NS entries of acc[]
    double4_t acc2[NS];
#pragma unroll
    for (int t = 0; t < NS; ++t) acc2[t] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int t = 0; t <= NH; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(x0[t][0], y0[0], acc[t], 0, 0, 1);
#pragma unroll
    for (int t = 0; t < NS; ++t) acc2[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(x1[S0 + t][0], y1[0], acc2[t], 0, 0, 1);
#pragma unroll
    for (int t = 0; t <= NH; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(x0[t][1], y0[1], acc[t], 0, 0, 1);
#pragma unroll
    for (int t = 0; t < NS; ++t) acc2[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(x1[S0 + t][1], y1[1], acc2[t], 0, 0, 1);
#pragma unroll
    for (int t = 0; t < S0; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(x1[t][0], y1[0], acc[t], 0, 0, 1);
#pragma unroll
    for (int t = 0; t < S0; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(x1[t][1], y1[1], acc[t], 0, 0, 1);
#pragma unroll
    for (int t = 0; t < NS; ++t) acc[S0 + t] = acc[S0 + t] + acc2[t];
  }
#pragma unroll
  for (int t = 0; t < NH; ++t) {
    double* im = Lim + (i0 + C2_EW * t) * C2_TSZ;
#pragma unroll
    for (int v = 0; v < 4; ++v) im[(g + 4 * v) * C2_TS + r] = acc[t][v];
  }
#pragma unroll
  for (int v = 0; v < 4; ++v) Dupd[(g + 4 * v) * C2_TS + r] = acc[NH][v];
}

// 16-byte global store with system scope (sc0 sc1): written through the XCD's L2, visible to the other XCDs once vmcnt has drained
// (`sc1` alone is NOT enough across XCDs on this part: the consumer then reads stale tiles - tried, every flavour of load)
__device__ __forceinline__ void c2_store_through_sys(double* p, dbl2_t v) {
  // (s_nop: a VALU write of the data registers of a 16-byte store needs two wait states behind it - the hazard recognizer does
  //  not see instructions inside an asm statement, and the compiler is free to reuse the registers at once: it did, for the address
  //  of the next store)
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
// the matching loads: two 16-byte pieces and the wait in ONE asm statement - the compiler does not know that an asm load
// completes later, with the wait in a statement of its own it is free to copy the destination registers in between
__device__ __forceinline__ void c2_load2_through_sys(const double* p0, const double* p1, dbl2_t& v0, dbl2_t& v1) {
  asm volatile(
      "global_load_dwordx4 %0, %2, off sc0 sc1\n\t"
      "global_load_dwordx4 %1, %3, off sc0 sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(v0), "=&v"(v1)
      : "v"(p0), "v"(p1)
      : "memory");
}

#define C2_WSYNC()                                           \
  do {                                                       \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   \
    __builtin_amdgcn_wave_barrier();                         \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");   \
  } while (0)

// Roles.  Waves 0..3 eliminate (VALU, DPP chains), waves 4..11 hold the tiles and run the trailing update (MFMA): the
// elimination of panel k + 1 overlaps the trailing update of step k on the other execution pipe of the same SIMDs.
// Since round 3 the LAST update of a column - panel k into column k + 1 - belongs to the elimination waves too (c2_last_update:
// their own MFMAs on their own tiles, in the row-major LDS image the tile waves published a step ahead), so the chain
// elimination -> update of the next column -> elimination never leaves them; the tile waves apply panel k to the columns
// from k + 2 on and publish column k + 2, complete up to that panel, at the end of the step.
// Hand-over through LDS counters (a wave's LDS traffic is in order, so data written before the increment is visible to whoever
// sees the increment):
//   cnt_col   += 1 per tile wave once its tiles of a column are in LDS (Dbuf + the images of L_ii behind it, free until step i)
//   cnt_used  += 1 per elimination wave once it has that column in registers: the same LDS words may take the next one
//   cnt_panel += 1 per elimination wave once its part of the panel is in LDS (tile waves, and the other elimination waves)
//   cnt_trail += 1 per tile wave at the end of a step: the panel buffer of that step may be overwritten two steps later
// (busy polling: with s_sleep 1 between two looks a hand-over was noticed ~60 cycles later on average, 1.3 % of the config-3 step)
// Every spin is bounded: a hand-over that never comes (it cannot, as long as every wave walks the same step sequence - a bad pivot
// does not change it, NaNs just flow through) raises the timeout word instead of hanging the CU until the watchdog; the caller
// turns it into a failed factorization.  ~100 cycles per look, 2^22 looks = 0.2 s.
static constexpr int C2_SPIN_LIMIT = 1 << 22;
__device__ __forceinline__ void c2_wait_ge(int* ctr, int target, int* tmo) {
  int spins = 0;
  while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) {
    asm volatile("s_nop 7");
    if (__builtin_expect(++spins > C2_SPIN_LIMIT, 0)) {
      __hip_atomic_store(tmo, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      break;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// The counter behind the data WITHOUT the wait of a release fence (round 6; until then `fence release` + atomic add: an
// `s_waitcnt lgkmcnt(0)` - one LDS round trip - in front of every hand-over, most of them on the critical chain, and a dozen
// instructions of wave-aggregation around the add).  The LDS executes a wave's operations in order and keeps no copies: once the
// add has executed, the wave's stores in front of it have, and a reader that has seen the increment reads them.  The compiler is held
// by the memory clobbers; its lgkmcnt bookkeeping does not know the add is in flight, which only makes its waits more conservative
// (the counter is in order for LDS operations).
__device__ __forceinline__ void c2_signal(int* ctr, int lane) {
  const unsigned a = (unsigned)(size_t)(__attribute__((address_space(3))) int*)ctr;
  asm volatile("" ::: "memory");
  if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"(a), "v"(1) : "memory");
  asm volatile("" ::: "memory");
}

// f(slot) for the slots lo..hi (wave-uniform bounds) of the statically indexed tile registers: one jump into the unrolled
// sequence instead of a test per slot (a skipped slot costs ~40 cycles of branch and instruction fetch)
template <int MAXSLOT, class F>
__device__ __forceinline__ void slot_range(int lo, int hi, F&& f) {
  if (lo > hi) return;
#define C2_CASE(S)                                                        \
  case S:                                                                 \
    if constexpr (S < MAXSLOT) f(std::integral_constant<int, S>{});       \
    if (hi <= S) break;                                                   \
    [[fallthrough]];
  switch (lo) {
    C2_CASE(0) C2_CASE(1) C2_CASE(2) C2_CASE(3) C2_CASE(4) C2_CASE(5) C2_CASE(6) C2_CASE(7) C2_CASE(8) C2_CASE(9) C2_CASE(10)
    C2_CASE(11) C2_CASE(12) C2_CASE(13) C2_CASE(14) C2_CASE(15) C2_CASE(16) C2_CASE(17) C2_CASE(18) C2_CASE(19) C2_CASE(20)
    C2_CASE(21) C2_CASE(22) C2_CASE(23)
    default: break;
  }
#undef C2_CASE
}

// LDS carve-up (doubles)
struct Chol2Lds {
  double* Dbuf;   // diagonal block of the next column as the tile waves publish it, row-major
  double* Dupd;   // ... with the last panel applied (every elimination wave writes the same values)
  double* PB;     // 2 x nt panel tiles
  double* Dsave;  // nt finished diagonal blocks L_kk (row-major, zero above the diagonal)
  double* zbuf;   // border row of the factor: z = L^-1 brow^T   [nt * 16]
  double* ybuf;   // back substitution result                     [nt * 16]
  double* pivs;   // pivots before the square root               [nt * 16]
  double* slots;  // [C2_TW][16] partial sums of the back substitution
  int* cnt;       // [0] cnt_col, [1] cnt_panel, [2] cnt_trail, [4] cnt_y, [5] exports confirmed, [6] spin timeout, [7] cnt_used
};
__host__ __device__ constexpr int chol2_lds_doubles(int nt) { return C2_TSZ * (2 + 3 * nt) + 3 * 16 * nt + C2_TW * 16 + 8; }
__device__ __forceinline__ Chol2Lds chol2_carve(double* lds, int nt) {
  Chol2Lds s;
  s.Dbuf = lds;
  s.Dupd = s.Dbuf + C2_TSZ;
  s.PB = s.Dupd + C2_TSZ;
  s.Dsave = s.PB + 2 * nt * C2_TSZ;
  s.zbuf = s.Dsave + nt * C2_TSZ;
  s.ybuf = s.zbuf + 16 * nt;
  s.pivs = s.ybuf + 16 * nt;
  s.slots = s.pivs + 16 * nt;
  s.cnt = reinterpret_cast<int*>(s.slots + C2_TW * 16);
  return s;
}

// Element (r, c) of the matrix that is factorized: the n x n input (+ I), the border row n, identity padding behind it.
template <class JOB>
__device__ __forceinline__ double c2_elem(const JOB& J, const double* __restrict__ A, int r, int c, int nb) {
  const int n = J.n;
  double x;
  if (r < n && c < n) {
    x = A[(size_t)r * J.ld + c];
    if (r == c && J.add_identity) x += 1.0;
  } else if (r == n && nb > n) {
    x = (c < n) ? J.brow[c] : (c == n ? 1e300 : 0.0);
  } else if (c == n && nb > n) {
    x = (r < n) ? J.brow[r] : 0.0;  // (upper part, never used)
  } else {
    x = (r == c) ? 1.0 : 0.0;
  }
  return x;
}

// tile slots of tile wave tw in tile column k (contiguous: the tiles are dealt out cyclically over the column-major list)
__device__ __forceinline__ void c2_col_slots(int k, int nt, int tw, int& lo, int& hi, int off = 0) {
  const int cs = k * nt - (k * (k - 1)) / 2 - off;  // list index of tile (k, k) (off = list index of the first owned tile)
  const int ce = cs + (nt - k) - 1;           // ... of tile (nt - 1, k)
  lo = (cs - tw + C2_TW - 1) / C2_TW;         // smallest s with s * C2_TW + tw >= cs   (cs >= 0, tw < C2_TW)
  hi = (ce - tw >= 0) ? (ce - tw) / C2_TW : -1;
  if (cs - tw < 0) lo = 0;
}

// Cycle stamps (diagnostics, OVP_PL_STAMPS): compiled in only under -DOVP_C2_STAMPS (tools/build_c2_stamps.sh builds that library
// beside the product one).  Even as `if (J.stamps && ...)` around nothing they are not free: ~10 tests per step and wave, each a
// scalar branch on a chain that is all latency - 1 us of the 78 per launch (measured round 6: five more of them, +1.0 us).
#ifdef OVP_C2_STAMPS
#define C2_STAMPS_ON 1
#define C2_STAMP(kk, i)                                                                              \
  do {                                                                                               \
    if (J.stamps && lane == 0) J.stamps[(kk) * 16 + (i)] = (long long)__builtin_readcyclecounter();  \
  } while (0)
#else
#define C2_STAMPS_ON 0
#define C2_STAMP(kk, i) \
  do {                  \
  } while (0)
#endif

// The factorization proper.  On return (tile waves): tile[] = final factor tiles (MFMA accumulator layout); S.Dsave = diagonal
// blocks, S.zbuf = border row (if any), S.pivs = pivots.  Every wave of the workgroup (C2_WAVES x 64 threads) must call it.
// ROLE 0 = elimination wave, 1 = tile wave: the kernel body is instantiated once per role and the role decided once, at the top of
// the kernel.  With one body and `if (wave < C2_EW)` inside it the register allocator sees the tile registers live on the
// elimination waves' paths too (they merge with the tile waves' behind every role-specific block) and spills them around the
// dependent chains: 376 spill instructions and 190 KB of scratch writes per launch that carried dead values.  In the ROLE 0
// instantiation the tile array does not exist (143 spill instructions, all on the tile waves' side; no measurable change in time).
//
// Column split (plane loop, J.split_h > 0): the workgroup owns the tile columns [cl, ch) only.  Part A (cl = 0, ch = h) takes the
// steps 0 .. h-1 on its trapezoid and EXPORTS the rows >= h of every finished panel to global memory (J.xbuf, one flag per step);
// part B (cl = h, ch = nt) imports them - its first h steps are "remote": the elimination waves copy the panel instead of
// computing it, the tile waves run the trailing update on the trailing triangle - and then factorizes that triangle itself.
// Part A's trailing updates shrink to its own columns (its steps become pivot-chain bound), part B's run on a CU of their own,
// and no workgroup holds more than ~2/3 of the tiles (n = 285: 126 of 171 - they fit the registers again).
// SPLIT = false compiles the ownership tests away (cl = 0, ch = nt): the single-workgroup kernel is the code it was before.
template <int MAXSLOT, int ROLE, bool SPLIT, class JOB>
__device__ __forceinline__ void chol2_factor(const JOB& J, const Chol2Lds& S, double4_t (&tile)[MAXSLOT], int (&ti)[MAXSLOT],
                                             int (&tj)[MAXSLOT], int& bad_out, const int cl_arg, const int ch_arg) {
  const int n = J.n;
  const int nb = J.brow ? n + 1 : n;  // bordered dimension
  const int nt = (nb + 15) >> 4;
  const int cl = SPLIT ? cl_arg : 0, ch = SPLIT ? ch_arg : nt;
  const int loff = cl * nt - (cl * (cl - 1)) / 2;                    // list index (full triangle) of the first owned tile
  const int ntiles = (ch * nt - (ch * (ch - 1)) / 2) - loff;         // owned tiles: columns cl .. ch-1, column-major
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane >> 4, lc = lane & 15;
  const int tb = n >> 4, rb = n & 15;  // tile row / row inside the tile of the border row
  // n a multiple of 16: the border row is alone in the last tile row.  Its tiles are panel tiles of every step (that is where z
  // comes from), but the last tile COLUMN holds nothing anybody reads (the corner of the border) - its step is not taken.
  const int nst = (nb > n && rb == 0) ? nt - 1 : nt;
  const int kend = ch < nst ? ch : nst;     // steps this workgroup takes part in
  const bool exporting = SPLIT && ch < nst;  // part A of a split factorization
  int* cnt_col = S.cnt;
  int* cnt_panel = S.cnt + 1;
  int* cnt_trail = S.cnt + 2;
  int* cnt_used = S.cnt + 7;
  if (tid < 8) S.cnt[tid] = 0;
  for (int i = tid; i < 16 * nt; i += C2_WAVES * 64) {
    S.zbuf[i] = 0.0;
    // the border row, staged once (the back substitution's y buffer is free until then): the patch of the border tiles in the
    // prologue then reads LDS instead of paying a global-memory round trip in front of column 0
    S.ybuf[i] = (nb > n && i < n) ? J.brow[i] : 0.0;
  }
  if constexpr (ROLE == 1)
    sfor<MAXSLOT>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      ti[s] = -1;
      tj[s] = -1;
      tile[s] = double4_t{0.0, 0.0, 0.0, 0.0};
    });
  __syncthreads();
  bool bad = false;

  if constexpr (ROLE == 0) {
    // =============================== elimination waves ===============================
    const int ew = wave, g = lr, r = lc;
    const double floor_eff = J.floor_scale ? J.piv_floor * (*J.floor_scale) : J.piv_floor;
    __builtin_amdgcn_s_setprio(3);  // the serial chain: its instructions go first, the tile waves fill the gaps
    for (int k = 0; k < kend; ++k) {
      double* pbk = S.PB + (k & 1) * nt * C2_TSZ;
      if (ew == 0) C2_STAMP(k, 0);
      if (k < cl) {
        // ---- remote step: panel k comes from part A (rows cl .. nt-1, row-major 16 x 16 tiles in J.xbuf) ----
        {
          int spins = 0;
          while (__hip_atomic_load(J.xflag + k, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != J.xseq) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 19)) {
              __hip_atomic_store(S.cnt + 6, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              break;
            }
          }
        }
        if (k >= 2) c2_wait_ge(cnt_trail, C2_TW * (k - 1), S.cnt + 6);  // the tile waves are done with panel k - 2 (same buffer)
        const double* xk = J.xbuf + (size_t)k * nt * 256;
        const int row = lane >> 2, c4 = (lane & 3) * 4;
        for (int i = cl + ew; i < nt; i += C2_EW) {
          const double* src = xk + (size_t)i * 256 + (c4 >> 1) * 32 + 2 * row;  // columns c4, c4+1 | c4+2, c4+3 of row `row`
          // `sc0 sc1` on both sides (stores in part A, loads here): the form that needs no L2 write-back / invalidate
          dbl2_t v0, v1;
          c2_load2_through_sys(src, src + 32, v0, v1);
          dbl2_t* dst = reinterpret_cast<dbl2_t*>(pbk + i * C2_TSZ + row * C2_TS + c4);
          dst[0] = v0;
          dst[1] = v1;
        }
        c2_signal(cnt_panel, lane);
        if (ew == 0) C2_STAMP(k, 3);
        continue;
      }
      double* pbp = S.PB + ((k + 1) & 1) * nt * C2_TSZ;  // panel k - 1
      // column k is in LDS (without panel k - 1: that one is applied here), panel k - 1 is complete (the other waves' tiles / the
      // import), and the tile waves are done with panel k - 2, whose buffer receives panel k: one look at the three counters per
      // round (three waits in a row were three LDS round trips, 700 cycles of a late step)
      {
        const int t_col = C2_TW * (k - cl + 1), t_panel = C2_EW * k, t_trail = C2_TW * (k - 1);
        int spins = 0;
        for (;;) {
          const int a = __hip_atomic_load(cnt_col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          const int b = __hip_atomic_load(cnt_panel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          const int c = __hip_atomic_load(cnt_trail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (C2_STAMPS_ON && J.stamps && spins == 0 && ew == 0 && lane == 0)  // diagnostics: what the step found on arrival
            J.stamps[k * 16 + 4] = (long long)(a - t_col) * 1000000 + (long long)(b - t_panel) * 1000 + (c - t_trail) + 500500500;
          if (a >= t_col && b >= t_panel && c >= t_trail) break;
          asm volatile("s_nop 7");
          if (__builtin_expect(++spins > C2_SPIN_LIMIT, 0)) {
            __hip_atomic_store(S.cnt + 6, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            break;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      }
      if (ew == 0) C2_STAMP(k, 1);
      bool flushed = false, signalled = false, used = false;
      // panel tiles k+1 .. nt-1 are dealt out over (elimination wave, DPP row): 16 per pass, a second pass only when the
      // column has 17 of them (bordered dimension 273..288, first column)
      for (int base = 0; base == 0 || k + 1 + base < nt; base += 4 * C2_EW) {
        const int my_i = k + 1 + base + ew + C2_EW * g;
        const bool has_p = my_i < nt;
        const bool first = (base == 0);
        if ((first && ew == 0) || k + 1 + base + ew < nt) {
          double d[16], p[16];
          // lanes without a panel tile of their own (DPP rows behind the end of the column) repeat the wave's first tile - same
          // values into the same LDS words - or, when the wave has none at all (the last step: no other wave is at work), write
          // into Dupd, which it has read by then: the panel stores of the elimination then need no lane mask
          const bool wave_has = k + 1 + base + ew < nt;  // (wave-uniform)
          const int src_i = has_p ? my_i : k + 1 + base + ew;
          dbl2_t* pw = reinterpret_cast<dbl2_t*>((wave_has ? pbk + src_i * C2_TSZ : S.Dupd) + r * C2_TS);
          // ---- the last update of the column, by the waves that eliminate it: tile (i, k) -= P_i P_k^T with panel k - 1 ----
          // The tile waves hand over column k with the panels up to k - 2 applied, a step ahead; what used to sit between two
          // eliminations - panel signal, a tile wave's look at it (behind whatever it was updating), its product, publication, the
          // elimination waves' look at that - is this wave's own product on its own tiles now: no hand-over, and the tile waves'
          // trailing update never holds up the chain.  The diagonal block is updated by every wave (same values into Dupd).
          if (k >= 1) {
            const int i0 = k + 1 + base + ew;  // (wave-uniform: tile rows of the four DPP rows are i0, i0 + 4, i0 + 8, i0 + 12)
            const int nh = i0 >= nt ? 0 : ((nt - i0 + C2_EW - 1) / C2_EW < 4 ? (nt - i0 + C2_EW - 1) / C2_EW : 4);
            const bool skip = C2_STAMPS_ON && (J.dbg & 2) != 0;  // (timing experiments: diagnostics build only)
            switch (nh) {
              case 0: c2_last_update<0>(S.Dbuf, S.Dupd, S.Dsave, pbp, k, i0, g, r, skip); break;
              case 1: c2_last_update<1>(S.Dbuf, S.Dupd, S.Dsave, pbp, k, i0, g, r, skip); break;
              case 2: c2_last_update<2>(S.Dbuf, S.Dupd, S.Dsave, pbp, k, i0, g, r, skip); break;
              case 3: c2_last_update<3>(S.Dbuf, S.Dupd, S.Dsave, pbp, k, i0, g, r, skip); break;
              default: c2_last_update<4>(S.Dbuf, S.Dupd, S.Dsave, pbp, k, i0, g, r, skip); break;
            }
            C2_WSYNC();  // (the rows are read back by other lanes of this wave)
          }
          {
            const dbl2_t* dr = reinterpret_cast<const dbl2_t*>((k >= 1 ? S.Dupd : S.Dbuf) + r * C2_TS);
            const dbl2_t* pr = reinterpret_cast<const dbl2_t*>(S.Dsave + (wave_has ? src_i : k) * C2_TSZ + r * C2_TS);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
              const dbl2_t dv = dr[q];
              d[2 * q] = dv[0];
              d[2 * q + 1] = dv[1];
              const dbl2_t pq = pr[q];
              p[2 * q] = wave_has ? pq[0] : 0.0;
              p[2 * q + 1] = wave_has ? pq[1] : 0.0;
            }
          }
          if (k + 1 + base + 4 * C2_EW >= nt) {
            // the column is in registers: its LDS image (Dbuf and the L_ii images behind k) may receive the next one
#pragma unroll
            for (int c = 0; c < 16; ++c) asm volatile("" : "+v"(d[c]), "+v"(p[c]));
            c2_signal(cnt_used, lane);
            used = true;
          }
          if (!(C2_STAMPS_ON && (J.dbg & 1))) {
            dbl2_t* pvw = reinterpret_cast<dbl2_t*>(S.pivs + 16 * k);
            // (the plane update, mode 1, never looks at its pivots: the stores are 2 us of its 75)
            dbl2_t* dw = reinterpret_cast<dbl2_t*>(S.Dsave + k * C2_TSZ + r * C2_TS);
            if (first && ew == 0) {  // (this wave also puts L_kk down)
              if (floor_eff > 0.0) bad = fused_elim16<true, true, true>(d, p, pvw, floor_eff, pw, dw, r) || bad;
              else if (J.mode == 1) bad = fused_elim16<false, false, true>(d, p, pvw, 0.0, pw, dw, r) || bad;
              else bad = fused_elim16<false, true, true>(d, p, pvw, 0.0, pw, dw, r) || bad;
            } else {
              if (floor_eff > 0.0) bad = fused_elim16<true, true, false>(d, p, pvw, floor_eff, pw, dw, r) || bad;
              else if (J.mode == 1) bad = fused_elim16<false, false, false>(d, p, pvw, 0.0, pw, dw, r) || bad;
              else bad = fused_elim16<false, true, false>(d, p, pvw, 0.0, pw, dw, r) || bad;
            }
          }
          if (ew == 0) C2_STAMP(k, 2);
          // the panel is out: the tile waves start on the next column while the rest of this step's results is put down
          if (k + 1 + base + 4 * C2_EW >= nt) {  // (last pass of the column)
            if (exporting && !flushed) {  // (wave-uniform)
              // the exports of the PREVIOUS step have long reached memory: confirming that here costs nothing, whereas waiting
              // for this step's stores (a memory round trip, 3 - 5 K cycles) would sit on the elimination chain
              asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
              c2_signal(S.cnt + 5, lane);
              flushed = true;
            }
            c2_signal(cnt_panel, lane);
            signalled = true;
          }
          if (has_p) {
            // (the border row z: its entries are taken from the finished tiles by the tile wave that owns them - sixteen stores
            //  under a one-lane mask here were in front of this wave's next step)
            if (exporting && my_i >= ch) {  // rows part B needs: the finished panel tile, row r of it from this lane

              // written THROUGH the L2 (agent scope): an agent-scope release fence here instead (write-back of the whole L2 +
              // wait) cost 3.3 - 5 K cycles per step on the elimination chain
              // tile layout in xbuf: element (r, 2q + e) at q * 32 + 2 r + e - the 16 lanes of a DPP row write 256 contiguous bytes
              // per instruction (whole lines; lane-major rows were 64 sixteen-byte pieces in 64 lines, 4 K cycles per step)
              double* xw = J.xbuf + ((size_t)k * nt + my_i) * 256 + 2 * r;
#pragma unroll
              for (int q = 0; q < 8; ++q) c2_store_through_sys(xw + 32 * q, dbl2_t{p[2 * q], p[2 * q + 1]});
            }
          }
          // (L_kk went into LDS inside the chain, in front of this wave's panel signal: whoever has seen panel k has it)
        }
      }
      if (!used) c2_signal(cnt_used, lane);
      if (!signalled) {  // a wave without work in this step
        if (exporting && !flushed) {  // (it still confirms its exports of the previous one)
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          c2_signal(S.cnt + 5, lane);
        }
        c2_signal(cnt_panel, lane);
      }
      if (exporting && ew == 0) {
        // in the gap in front of the next column
        if (k > 0) {
          c2_wait_ge(S.cnt + 5, C2_EW * (k + 1), S.cnt + 6);  // every wave has seen its exports of step k - 1 complete
          if (lane == 0) __hip_atomic_store(J.xflag + k - 1, J.xseq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      if (ew == 0) C2_STAMP(k, 3);
    }
    if (exporting) {  // the last step's exports: the one memory round trip that is waited for
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      c2_signal(S.cnt + 5, lane);
      if (ew == 0) {
        c2_wait_ge(S.cnt + 5, C2_EW * (kend + 1), S.cnt + 6);
        if (lane == 0) __hip_atomic_store(J.xflag + kend - 1, J.xseq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    __builtin_amdgcn_s_setprio(0);
  } else {
    // ================================== tile waves ===================================
    const int tw = wave - C2_EW;
    if (tw == 0) C2_STAMP(1, 13);
    // ---- load: tile index idx = slot * C2_TW + tw, column-major over the lower tile triangle ----
    // Column 0 first (its tiles are the first nt of the list = slots 0..2 of every wave): loaded, patched and handed to the
    // elimination waves before the other ~120 tiles are requested - those then stream in (the CU's memory pipe at 8-byte loads,
    // ~9 K cycles) while column 0 is being eliminated, instead of in front of it.
    const int s_last = (ntiles - 1 - tw >= 0) ? (ntiles - 1 - tw) / C2_TW : -1;
    auto put_rowmajor = [&](double* buf, const double4_t& t) {
#pragma unroll
      for (int v = 0; v < 4; ++v) buf[(lr + 4 * v) * C2_TS + lc] = t[v];
    };
    auto get_acc = [&](const double* buf) {
      double4_t t;
#pragma unroll
      for (int v = 0; v < 4; ++v) t[v] = buf[(lr + 4 * v) * C2_TS + lc];
      return t;
    };
    // the diagonal tiles (+ I) and the tile rows that hold the border row / the identity padding
    auto patch_slot = [&](auto sc) {
      constexpr int s = decltype(sc)::value;
      const int i = ti[s], j = tj[s];
      if (i >= 0 && (i >= (n >> 4) || i == j)) {
        const int c = 16 * j + lc;
        const double brow_c = (nb > n && i == tb) ? S.ybuf[c < n ? c : n - 1] : 0.0;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int r = 16 * i + lr + 4 * v;
          double pad = (r == c) ? 1.0 : 0.0;
          if (nb > n && r == n) pad = (c < n) ? brow_c : (c == n ? 1e300 : 0.0);
          double x = tile[s][v];
          if (J.flip && J.boost && r == c && r < n && r >= n - J.boost_n) {  // state column n - 1 - r < boost_n
            const double add = x * J.boost_rel;
            J.boost[n - 1 - r] = add;
            x += add;
          }
          tile[s][v] = (r < n && c < n) ? ((r == c && J.add_identity) ? x + 1.0 : x) : pad;
        }
      }
    };
    {
      const double* Abase = J.A + (J.sel ? (size_t)((*J.sel) ^ J.sel_xor) * J.sel_stride : (size_t)0);
      int jj = cl, cstart = 0;
      constexpr int SC0 = MAXSLOT < 3 ? MAXSLOT : 3;  // slots that can hold tiles of column 0 (nt <= 18 < 3 * C2_TW)
      auto load_slot = [&](auto sc) {
        constexpr int s = decltype(sc)::value;
        const int idx = s * C2_TW + tw;
        int i = -1, j = -1;
        if (idx < ntiles) {
          while (cstart + (nt - jj) <= idx) {
            cstart += nt - jj;
            ++jj;
          }
          j = jj;
          i = jj + (idx - cstart);
        }
        i = __builtin_amdgcn_readfirstlane(i);
        j = __builtin_amdgcn_readfirstlane(j);
        ti[s] = i;
        tj[s] = j;
        // raw loads from clamped addresses, nothing uses them yet - all requested tiles of the wave are in flight together
        double4_t t = {0.0, 0.0, 0.0, 0.0};
        if (i >= 0) {
          const int c = 16 * j + lc;
          const int cc = c < n ? c : n - 1;
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int r = 16 * i + lr + 4 * v;
            const int rc = r < n ? r : n - 1;
            t[v] = J.flip ? Abase[(size_t)(n - 1 - rc) * J.ld + (n - 1 - cc)] : Abase[(size_t)rc * J.ld + cc];
          }
        }
        tile[s] = t;
      };
      sfor<SC0>(load_slot);
      if (tw == 0) C2_STAMP(0, 13);
      sfor<SC0>(patch_slot);
      if (tw == 0) C2_STAMP(0, 14);
      if (cl == 0) {
        // column 0: diagonal block and panel tiles go to LDS
        sfor<SC0>([&](auto sc) {
          constexpr int s = decltype(sc)::value;
          if (ti[s] >= 0 && tj[s] == 0) {
            if (ti[s] == 0) put_rowmajor(S.Dbuf, tile[s]);
            else put_rowmajor(S.Dsave + ti[s] * C2_TSZ, tile[s]);
          }
        });
        c2_signal(cnt_col, lane);
      }
      if (tw == 0) C2_STAMP(0, 15);
      // the rest of the triangle
      sfor<MAXSLOT - SC0>([&](auto sc) { load_slot(std::integral_constant<int, SC0 + decltype(sc)::value>{}); });
      sfor<MAXSLOT - SC0>([&](auto sc) { patch_slot(std::integral_constant<int, SC0 + decltype(sc)::value>{}); });
      if (cl == 0 && 1 < kend) {
        // column 1 receives its only panel from the elimination waves: published as loaded, once they have taken column 0.
        // (Earlier - in front of the patch above, which waits for the last load of the batch - does not pay: step 2 of the elimination
        // waves needs the tile waves' step 0 finished, the buffer of panel 0 becomes that of panel 2, and that is ~35 K cycles
        // away whatever the order: loads, then 105 tiles x 4 MFMAs.)
        int lo1 = 0, hi1 = -1;
        c2_col_slots(1, nt, tw, lo1, hi1, loff);
        c2_wait_ge(cnt_used, C2_EW, S.cnt + 6);
        slot_range<MAXSLOT>(lo1, hi1, [&](auto sc) {
          constexpr int s = decltype(sc)::value;
          if (ti[s] == 1) put_rowmajor(S.Dbuf, tile[s]);
          else put_rowmajor(S.Dsave + ti[s] * C2_TSZ, tile[s]);
        });
        c2_signal(cnt_col, lane);
      }
    }
    // Step k of a tile wave: panel k goes into every owned tile behind column k + 1 - column k + 1 itself gets it from the
    // elimination waves, who own the chain - and column k + 2, then complete up to that panel, is published for them: a step
    // ahead of its use, so nothing here is waited for by the chain (the publication only has to follow the elimination waves'
    // "column k + 1 taken": it goes into the same LDS words - Dbuf and the images of L_ii, i >= k + 3, free until step i).
    // (tile row, tile column) of a slot as ONE scalar inside the step loop, unpacked where it is used from a per-step opaque copy:
    // with ti[] / tj[] the compiler hoists every slot's `ti * tile size`, `tj * tile size` out of the step loop - 45 live scalars for
    // fifteen slots, most of them spilled to VGPR lanes and fetched back with v_readlane (VALU, behind the other waves' MFMAs)
    int tij[MAXSLOT];
    sfor<MAXSLOT>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      tij[s] = (ti[s] & 0xff) | ((tj[s] & 0xff) << 8);
    });
    for (int k = 0; k < kend; ++k) {
      double* pbk = S.PB + (k & 1) * nt * C2_TSZ;
      int tq[MAXSLOT];
      sfor<MAXSLOT>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        tq[s] = tij[s];
        asm volatile("" : "+s"(tq[s]));
      });
      auto TI = [&](auto sc) { return (tq[decltype(sc)::value] << 24) >> 24; };  // (signed bytes: -1 = no tile)
      auto TJ = [&](auto sc) { return (tq[decltype(sc)::value] << 16) >> 24; };
      int lo = 0, hi = -1, lo2 = 0, hi2 = -1;
      const bool pub = (k + 2 >= cl) && (k + 2 < kend);  // column k + 2 is this workgroup's to publish
      // per-step opaque copies: otherwise the LDS address arithmetic of every slot is hoisted out of the step loop and spills
      int lc_k = lc, lr_k = lr;
      asm volatile("" : "+v"(lc_k), "+v"(lr_k));
      int s_last_k = s_last;  // (opaque too: the fifteen `slot <= s_last` flags are loop invariants otherwise, spilled to VGPR lanes)
      asm volatile("" : "+s"(s_last_k));
      int off_k = 8 * (lc_k * C2_TS + 4 * lr_k);  // (the lane's operand offset of the trailing update in bytes, see c2_mfma_xyT)
      asm volatile("" : "+v"(off_k));
      if (k >= cl) c2_col_slots(k, nt, tw, lo, hi, loff);
      // the lane's place in a tile image in the ACCUMULATOR layout (rows lr + 4 v of column lc), in bytes, once per step: the four
      // elements are then one address add and immediate offsets (as `(lr_k + 4 v) * C2_TS + lc_k` the address arithmetic was rebuilt
      // at every node of the slot dispatch - VALU instructions that queue behind the other waves' MFMAs)
      int acc_off_k = 8 * (lr_k * C2_TS + lc_k);
      asm volatile("" : "+v"(acc_off_k));
      auto put_rowmajor_k = [&](double* buf, const double4_t& t) {
        double* b = reinterpret_cast<double*>(reinterpret_cast<char*>(buf) + acc_off_k);
#pragma unroll
        for (int v = 0; v < 4; ++v) b[4 * v * C2_TS] = t[v];
      };
      auto get_acc_k = [&](const double* buf) {
        const double* b = reinterpret_cast<const double*>(reinterpret_cast<const char*>(buf) + acc_off_k);
        double4_t t;
#pragma unroll
        for (int v = 0; v < 4; ++v) t[v] = b[4 * v * C2_TS];
        return t;
      };
      if (k + 2 >= cl) c2_col_slots(k + 2, nt, tw, lo2, hi2, loff);  // (lo2 = first slot behind column k + 1; 0 = every owned tile)
      if (hi2 > s_last_k) hi2 = s_last_k;  // (a column of the other workgroup: nothing of it is in the list)
      if (tw == 0) C2_STAMP(k, 8);
      c2_wait_ge(cnt_panel, C2_EW * (k + 1), S.cnt + 6);  // panel k is in LDS
      if (tw == 0) C2_STAMP(k, 9);
      if (tw == C2_TW - 1) C2_STAMP(k, 7);
      // ---- the trailing update (column k + 2 first in the list) ----
      slot_range<MAXSLOT>(lo2, s_last_k, [&](auto sc) {
        constexpr int s = decltype(sc)::value;
        if (!(C2_STAMPS_ON && (J.dbg & 2))) tile[s] = c2_mfma_xyT(pbk + TI(sc) * C2_TSZ, pbk + TJ(sc) * C2_TSZ, tile[s], off_k);
      });
      if (tw == 0) C2_STAMP(k, 10);
      if (tw == C2_TW - 1) C2_STAMP(k, 5);  // (the last tile wave - the youngest on its SIMD, the one the step waits for)
      // own tiles of column k take their final values (the panel buffer lives until step k + 2; L_kk is put down behind the
      // elimination waves' signal, its owner waits for a counter of its own)
      slot_range<MAXSLOT>(lo, hi, [&](auto sc) {
        constexpr int s = decltype(sc)::value;
        const int ti_s = TI(sc);
        tile[s] = get_acc_k((ti_s == k) ? S.Dsave + k * C2_TSZ : pbk + ti_s * C2_TSZ);
        // the border row of the factor, z = L^-1 brow^T: row rb of the tiles of tile row tb (columns behind the border are not z)
        if (nb > n && ti_s == tb) {  // (wave-uniform: a branch, not a lane mask evaluated for every tile)
          asm volatile("");
          if (lr_k == (rb & 3)) {
            const int col = 16 * k + lc_k;
            if (col < n) S.zbuf[col] = tile[s][rb >> 2];
          }
        }
      });
      // column k + 2 goes out LAST: the elimination waves need it a whole step of theirs from now, and they have taken column
      // k + 1 out of the same LDS words by now (publishing right behind its update - the tile wave idled 2 - 10 K cycles for that
      // in the first steps, where the elimination waves are themselves held up by the tile waves' previous update)
      if (tw == 0 && k >= 2) C2_STAMP(k, 13);  // (rows 0 and 1 carry the prologue's stamps there)
      if (pub) {
        c2_wait_ge(cnt_used, C2_EW * (k + 2 - cl), S.cnt + 6);
        if (tw == 0 && k >= 2) C2_STAMP(k, 14);
        slot_range<MAXSLOT>(lo2, hi2, [&](auto sc) {
          constexpr int s = decltype(sc)::value;
          const int ti_s = TI(sc);
          if (ti_s == k + 2) put_rowmajor_k(S.Dbuf, tile[s]);
          else put_rowmajor_k(S.Dsave + ti_s * C2_TSZ, tile[s]);
        });
        c2_signal(cnt_col, lane);
      }
      if (tw == 0) C2_STAMP(k, 11);
      c2_signal(cnt_trail, lane);  // done with panel k
      if (tw == 0) C2_STAMP(k, 12);
      if (tw == C2_TW - 1) C2_STAMP(k, 6);
    }
    sfor<MAXSLOT>([&](auto sc) {  // (for the callers: the arrays need not live across the loop)
      constexpr int s = decltype(sc)::value;
      ti[s] = (tij[s] << 24) >> 24;
      tj[s] = (tij[s] << 16) >> 24;
    });
  }
  bad_out = (bad && !(J.piv_floor > 0.0)) ? 1 : 0;
}

// Inverses of the diagonal blocks L_kk, k in [k_lo, k_hi), in place in S.Dsave - elimination waves only (the 16 DPP rows of
// the four waves take one block each).  Lane r holds row r of L_kk (one batch of 16-byte LDS reads); lane c builds column c of
// X = L_kk^-1 by forward substitution, x_i = (delta_ic - sum_{j<i} L_ij x_j) / L_ii, with L_ij taken from lane i INSIDE the FMA
// (DPP row broadcast) - no LDS traffic and one division per lane on the dependent chain.
__device__ __forceinline__ void c2_invert_diag_blocks(const Chol2Lds& S, int k_lo, int k_hi, int wave, int lr, int lc) {
  for (int k = k_lo + wave * 4 + lr; k < k_hi; k += 4 * C2_EW) {
      double* blk = S.Dsave + k * C2_TSZ;
      // lane r holds row r of L_kk (one batch of 16-byte LDS reads); lane c builds column c of X = L_kk^-1 by forward
      // substitution, x_i = (delta_ic - sum_{j<i} L_ij x_j) / L_ii, with L_ij taken from lane i INSIDE the FMA (DPP row
      // broadcast) - no LDS traffic and one division per lane on the dependent chain
      double l[16];
      {
        const dbl2_t* lrow = reinterpret_cast<const dbl2_t*>(blk + lc * C2_TS);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const dbl2_t v = lrow[q];
          l[2 * q] = v[0];
          l[2 * q + 1] = v[1];
        }
      }
      double dmine = 0.0;
      sfor<16>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if (lc == i) dmine = l[i];
      });
      const double rmine = dmine != 0.0 ? 1.0 / dmine : 0.0;
      double x[16];
      sfor<16>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        double s0 = (i == lc) ? 1.0 : 0.0, s1 = 0.0;
        sfor<i>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          double nx = -x[j];
          if constexpr ((j & 1) == 0) fmac_bcast_nop<i>(s0, l[j], nx);  // s += L_ij (lane i's l[j]) * (-x_j)
          else fmac_bcast_nop<i>(s1, l[j], nx);
        });
        x[i] = (s0 + s1) * bcast_row<i>(rmine);
      });
      C2_WSYNC();  // every lane of the row has read L_kk
#pragma unroll
      for (int i = 0; i < 16; ++i) blk[i * C2_TS + lc] = x[i];
    }
}

// y = L^-T z on the n x n part of the factor held in tile[] (entries at or behind the border row are zero); nt = tile rows of that
// part, ceil(n / 16).
//
// Sixteen dependent block steps; what matters is the length of one step, so the chain  y_k = L_kk^-T (z_k - sum_{i>k} L_ik^T y_i)
// runs inside ONE wave with nothing but DPP-broadcast FMAs on it:
//   * preparation (all waves, once): the tile waves drop their sub-diagonal tiles L_(k+1)k into the panel buffers (free after the
//     factorization); the 16 DPP rows of the elimination waves invert one diagonal block each, lane c solving L x = e_c - the
//     inverse replaces L_kk in S.Dsave.  The diagonal solve of a step becomes a product with L_kk^-T instead of a 16-column chain.
//   * wave 0, step k:  v = z_k - S_k - L_(k+1)k^T y_(k+1),  y_k = L_kk^-T v : two 16 x 16 products as broadcast-in-FMA chains,
//     operands prefetched from LDS a step ahead.
//   * tile waves, one step AHEAD of the chain: S_k = sum_{i >= k+2} L_ik^T y_i from their register tiles (needs y_(k+2), which
//     was published a whole step earlier), one partial vector per wave, summed by wave 0 in a fixed order.
// Hand-over through two LDS counters; no workgroup barrier inside the recurrence (the first version had two per step and a
// 16-column substitution chain in between: 29 us at 16 tile columns).
template <int MAXSLOT, int ROLE>
__device__ __forceinline__ void chol2_backsolve(const Chol2Lds& S, int n, int nt, const double4_t (&tile)[MAXSLOT],
                                                const int (&ti)[MAXSLOT], const int (&tj)[MAXSLOT], long long* stamps = nullptr,
                                                int k_hi = -1, int k_lo = 0) {
  // Split factorization: this workgroup holds the tile columns [k_lo, k_hi) and runs that part of the recurrence; the y blocks
  // k_hi .. nt-1 were computed by the other part and are in S.ybuf already.
  if (k_hi < 0) k_hi = nt;
#define BS_STAMP(kk, i)                                                                       \
  do {                                                                                        \
    if (C2_STAMPS_ON && stamps && lane == 0) stamps[(kk) * 16 + (i)] = (long long)__builtin_readcyclecounter(); \
  } while (0)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane >> 4, lc = lane & 15;
  double* SD = S.PB;                        // [nt] sub-diagonal tiles, row-major, pitch C2_TS
  double* SL = S.PB + (size_t)nt * C2_TSZ;  // [nt][C2_TW][16] partial sums S_k
  int* cnt_y = S.cnt + 4;
  int* cnt_s = reinterpret_cast<int*>(S.Dbuf);  // one counter per column (the diagonal-block buffer is free): a tile wave with an
                                                // empty column may run ahead of the others, a single running count would not
                                                // say WHOSE partial sums are in
  BS_STAMP(nt, 3);
  if (tid == 0) *cnt_y = nt - k_hi;  // y blocks known from the start
  if (tid < 32) cnt_s[tid] = 0;
  // ---- preparation ----
  if constexpr (ROLE == 1) {
    sfor<MAXSLOT>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      if (ti[s] >= 0 && ti[s] == tj[s] + 1 && ti[s] < nt) {  // (nt = tile rows of the n x n part: a border-only tile row is not one)
        double* buf = SD + tj[s] * C2_TSZ;
#pragma unroll
        for (int v = 0; v < 4; ++v) buf[(lr + 4 * v) * C2_TS + lc] = tile[s][v];
      }
    });
  } else {
    c2_invert_diag_blocks(S, k_lo, k_hi, wave, lr, lc);
  }
  if (wave == 1)  // tile slot nt - 1 of SD has no sub-diagonal tile: the first step multiplies it by y = 0
    for (int e = lane; e < C2_TSZ; e += 64) SD[(nt - 1) * C2_TSZ + e] = 0.0;
  if (wave == 2 || wave == 3)  // partial sums of the waves that own no tile of a column stay zero
    for (int e = (wave - 2) * 64 + lane; e < nt * C2_TW * 16; e += 128) SL[e] = 0.0;
  BS_STAMP(nt, 4);
  __syncthreads();
  BS_STAMP(nt, 5);
  if constexpr (ROLE == 0) {
  if (wave == 0) {
    // ---- the chain ----
    const int r = lc;
    double y = k_hi < nt ? S.ybuf[16 * k_hi + r] : 0.0;
    double sd[16], di[16];
    // operands of a step are loaded into the registers the previous step has just finished with: the sub-diagonal tile right
    // after the first product, the inverse block right after the second (a second register set for a full prefetch made the
    // kernel spill inside the factorization)
    // (unconditional loads from a clamped tile index: a select per element turned into a branch per element)
    auto load_sd = [&](int k) {
      const double* sblk = SD + (k >= 0 ? k : 0) * C2_TSZ + r;
#pragma unroll
      for (int c = 0; c < 16; ++c) sd[c] = sblk[c * C2_TS];
    };
    auto load_di = [&](int k) {
      const double* dblk = S.Dsave + (k >= 0 ? k : 0) * C2_TSZ + r;
#pragma unroll
      for (int c = 0; c < 16; ++c) di[c] = dblk[c * C2_TS];
    };
    load_sd(k_hi - 1);
    load_di(k_hi - 1);
    for (int k = k_hi - 1; k >= k_lo; --k) {
      BS_STAMP(k, 4);
      // right-hand side, counter and partial sums in one batch of LDS reads in front of the first product, which does not need
      // them (a wave's LDS operations complete in order: sums read behind a counter that shows everybody are at least that new)
      static_assert(C2_TW == 8, "eight partial vectors");
      const int kc = k <= nt - 3 ? k : 0;
      const double* sl = SL + ((size_t)kc * C2_TW) * 16 + r;
      int got = __hip_atomic_load(cnt_s + kc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      asm volatile("" ::: "memory");  // compiler only: the sums are requested behind the counter
      double p0 = sl[0], p1 = sl[16], p2 = sl[32], p3 = sl[48], p4 = sl[64], p5 = sl[80], p6 = sl[96], p7 = sl[112];
      double v = S.zbuf[16 * k + r];
      // t = L_(k+1)k^T y_(k+1): needs only the previous step's y
      double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
      fmac_bcast_nop<0>(t0, y, sd[0]);
      fmac_bcast_nop<1>(t1, y, sd[1]);
      fmac_bcast_nop<2>(t2, y, sd[2]);
      fmac_bcast_nop<3>(t3, y, sd[3]);
      fmac_bcast<4>(t0, y, sd[4]);
      fmac_bcast<5>(t1, y, sd[5]);
      fmac_bcast<6>(t2, y, sd[6]);
      fmac_bcast<7>(t3, y, sd[7]);
      fmac_bcast<8>(t0, y, sd[8]);
      fmac_bcast<9>(t1, y, sd[9]);
      fmac_bcast<10>(t2, y, sd[10]);
      fmac_bcast<11>(t3, y, sd[11]);
      fmac_bcast<12>(t0, y, sd[12]);
      fmac_bcast<13>(t1, y, sd[13]);
      fmac_bcast<14>(t2, y, sd[14]);
      fmac_bcast<15>(t3, y, sd[15]);
      double tsum = (t0 + t1) + (t2 + t3);
      asm volatile("" : "+v"(tsum));  // the products above are finished before sd is reloaded
      load_sd(k - 1);
      BS_STAMP(k, 5);
      if (k <= nt - 3) {
        // column k has nt - k - 2 tiles below the sub-diagonal, dealt out cyclically: that many waves (at most all) report
        const int need = (nt - k - 2 < C2_TW) ? nt - k - 2 : C2_TW;
        int spins = 0;
        while (got < need) {  // (the first read was issued in front of the first product)
          asm volatile("s_nop 7");
          if (__builtin_expect(++spins > C2_SPIN_LIMIT, 0)) {
            __hip_atomic_store(S.cnt + 6, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            break;
          }
          got = __hip_atomic_load(cnt_s + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          asm volatile("" ::: "memory");
          p0 = sl[0], p1 = sl[16], p2 = sl[32], p3 = sl[48], p4 = sl[64], p5 = sl[80], p6 = sl[96], p7 = sl[112];
          asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7));
        }
        BS_STAMP(k, 6);
        v -= ((p0 + p1) + (p2 + p3)) + ((p4 + p5) + (p6 + p7));
      }
      v -= tsum;
      if (16 * k + r >= n) v = 0.0;
      double y0 = 0.0, y1 = 0.0, y2 = 0.0, y3 = 0.0;
      fmac_bcast_nop<0>(y0, v, di[0]);
      fmac_bcast_nop<1>(y1, v, di[1]);
      fmac_bcast_nop<2>(y2, v, di[2]);
      fmac_bcast_nop<3>(y3, v, di[3]);
      fmac_bcast<4>(y0, v, di[4]);
      fmac_bcast<5>(y1, v, di[5]);
      fmac_bcast<6>(y2, v, di[6]);
      fmac_bcast<7>(y3, v, di[7]);
      fmac_bcast<8>(y0, v, di[8]);
      fmac_bcast<9>(y1, v, di[9]);
      fmac_bcast<10>(y2, v, di[10]);
      fmac_bcast<11>(y3, v, di[11]);
      fmac_bcast<12>(y0, v, di[12]);
      fmac_bcast<13>(y1, v, di[13]);
      fmac_bcast<14>(y2, v, di[14]);
      fmac_bcast<15>(y3, v, di[15]);
      y = (y0 + y1) + (y2 + y3);
      if (16 * k + r >= n) y = 0.0;
      if (lane < 16) S.ybuf[16 * k + r] = y;
      c2_signal(cnt_y, lane);
      load_di(k - 1);
      BS_STAMP(k, 7);
    }
  }
  } else {
    // ---- partial sums, ahead of the chain ----
    // ONE static pass over the wave's tile registers in descending list order = columns from right to left, rows from the
    // bottom up inside a column - exactly the order in which the y blocks they need are published: the last tile of column k a
    // wave can own is (k + 2, k), so when y_(k+2) arrives one product, one reduction and the hand-over are left.  (Dispatching
    // into the unrolled tile sequence once per column, as the factorization does, cost ~17 scalar branches per column here.)
    const int tw = wave - C2_EW;
    int cur = nt;       // column whose partial sum is being accumulated (nt = none yet)
    int pub = nt - k_hi;  // y blocks known to be published
    double part = 0.0;
    bool any = false;   // the wave owns a tile (i >= cur + 2) of column cur
    auto flush_to = [&](int jnew) {  // closes column cur (a wave without tiles in a column says nothing: wave 0 knows how many
                                     // waves own tiles of it, the partial sums of the others were zeroed above)
      if (any && cur <= nt - 3) {
        const double pv = rows_sum_f64(part);
        if (lane < 16) SL[((size_t)cur * C2_TW + tw) * 16 + lane] = pv;
        c2_signal(cnt_s + cur, lane);
      }
      cur = jnew;
      part = 0.0;
      any = false;
    };
    sfor<MAXSLOT>([&](auto sc) {
      constexpr int s = MAXSLOT - 1 - decltype(sc)::value;
      const int i = ti[s], j = tj[s];
      if (i >= 0 && i < nt) {
        if (j != cur) flush_to(j);
        if (i >= j + 2) {
          if (pub < nt - i) {
            c2_wait_ge(cnt_y, nt - i, S.cnt + 6);
            pub = nt - i;
          }
#pragma unroll
          for (int v = 0; v < 4; ++v) part = fma(tile[s][v], S.ybuf[16 * i + lr + 4 * v], part);
          any = true;
        }
      }
    });
    flush_to(-1);
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------------------------------------------------
// tile-packed copy of a factor tile for k_fwdsub: tile index over the n x n part (nt_n tile rows), column-major inside the tile;
// rows / columns at or behind the border row are replaced by the identity
__device__ __forceinline__ void c2_pack_tile(double* __restrict__ Lpack, const double4_t& t, int i, int j, int n, int lr, int lc,
                                             int n_layout = 0) {
  const int ntn = ((n_layout > 0 ? n_layout : n) + 15) >> 4;  // the packed layout may belong to a larger matrix (leading-block factor)
  const int tidx = j * ntn - (j * (j - 1)) / 2 + (i - j);
  double* pk = Lpack + (size_t)tidx * 256;
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int row = lr + 4 * v, col = lc;
    const int gr = 16 * i + row, gc = 16 * j + col;
    double x = t[v];
    if (gr >= n || gc >= n) x = (gr == gc) ? 1.0 : 0.0;
    if (i == j && col > row) x = 0.0;
    pk[col * 16 + row] = x;
  }
}

// offset of the third kernel argument (k_chol2(Chol2Job, Chol2Job, PlaneSolve)) in the kernel-argument segment
static constexpr size_t C2_PS_KERNARG_OFFSET = (2 * sizeof(Chol2Job) + alignof(PlaneSolve) - 1) / alignof(PlaneSolve) * alignof(PlaneSolve);
static_assert(sizeof(Chol2Job) % alignof(Chol2Job) == 0 && alignof(Chol2Job) == 8 && alignof(PlaneSolve) == 8, "kernel-argument layout");

struct Chol2Shared {  // workgroup variables both role instantiations of the body see
  int bad, ok;
  double zz;
};

template <int MAXSLOT, int ROLE, bool SPLIT>
__device__ __forceinline__ void chol2_body(const Chol2Job& J0_, const PlaneSolve& ps, double* lds, Chol2Shared& sh, const int joff) {
  // The job is read from the kernel-argument segment directly (joff: where it sits there): as `blockIdx.x == 1 ? J1 : J0` every field
  // was two scalar loads and a select, all of them fetched at the kernel's entry.
  typedef const Chol2Job __attribute__((address_space(4))) Chol2JobK;
  typedef const char __attribute__((address_space(4))) KernargByte;
  Chol2JobK& Jf = *(Chol2JobK*)((KernargByte*)__builtin_amdgcn_kernarg_segment_ptr() + joff);
  const int n = Jf.n;
  const int nb = Jf.brow ? n + 1 : n;
  const int nt = (nb + 15) >> 4;
  const Chol2Lds S = chol2_carve(lds, nt);
  constexpr int NS = ROLE == 1 ? MAXSLOT : 1;  // tile registers exist on the tile waves only
  // cond = {have, want} on the device: the factor this launch would produce is already there when the two agree (the last
  // accepted plane of a plane loop left it behind)
  if (Jf.skip_cond && Jf.skip_cond[0] != 0 && Jf.skip_cond[0] == Jf.skip_cond[1]) return;
  double4_t tile[NS];
  int ti[NS], tj[NS];
  int bad = 0;
  int& sh_bad = sh.bad;
  int& sh_ok = sh.ok;
  double& sh_zz = sh.zz;
  if (threadIdx.x == 0) sh_bad = 0;
  // split factorization of the plane update (see chol2_factor): block 0 = part A (tile columns < h), block 2 = part B
  const int part = (SPLIT && Jf.mode == 1 && Jf.split_h > 0) ? (blockIdx.x == 2 ? 1 : 0) : -1;
  const int c_lo = part == 1 ? Jf.split_h : 0, c_hi = part == 0 ? Jf.split_h : nt;
#if C2_STAMPS_ON
  Chol2Job Jl = J0_;
  if (part == 1 && Jl.stamps) Jl.stamps += 16 * 32;  // diagnostics: part B stamps into the second half
  const Chol2Job& J = Jl;
  chol2_factor<NS, ROLE, SPLIT, Chol2Job>(J, S, tile, ti, tj, bad, c_lo, c_hi);
#else
  chol2_factor<NS, ROLE, SPLIT, Chol2JobK>(Jf, S, tile, ti, tj, bad, c_lo, c_hi);
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane >> 4, lc = lane & 15;
  if (bad && lane == 0) atomicOr(&sh_bad, 1);  // a wave only sees the pivots of the steps it took part in
  __syncthreads();
  if (tid == 0 && S.cnt[6]) atomicOr(&sh_bad, 2);  // a hand-over timed out (c2_wait_ge)
  __syncthreads();
  bad = sh_bad;
  // From here on the job and the plane's parameters are read AFRESH from the kernel-argument segment (scalar loads through pointers
  // the compiler cannot see through): as members of the by-value arguments every field the code behind the factorization uses - some
  // forty pointers and counts - was fetched at the kernel's entry and carried through the factorization in scalar registers, i.e. in
  // spill lanes of vector registers written and read back (v_writelane / v_readlane, VALU instructions) around its loops.
  typedef const PlaneSolve __attribute__((address_space(4))) PlaneSolveK;
  PlaneSolveK* pk = (PlaneSolveK*)((KernargByte*)__builtin_amdgcn_kernarg_segment_ptr() + C2_PS_KERNARG_OFFSET);
  Chol2JobK* jk = (Chol2JobK*)((KernargByte*)__builtin_amdgcn_kernarg_segment_ptr() + joff);
  asm volatile("" : "+s"(pk), "+s"(jk));
  PlaneSolveK& pq = *pk;
  Chol2JobK& jq = *jk;
  if (bad && tid == 0 && jq.flag) atomicOr(jq.flag, bad);

  if (jq.mode == 0) {
    // ---- outputs of a plain factorization ----
    if (jq.z_out)
      for (int i = tid; i < n; i += C2_WAVES * 64) jq.z_out[i] = S.zbuf[i];
    if (jq.y_out) {  // diagnostics: y = L^-T z
      chol2_backsolve<NS, ROLE>(S, n, (n + 15) >> 4, tile, ti, tj, jq.stamps);
      for (int i = tid; i < n; i += C2_WAVES * 64) jq.y_out[i] = S.ybuf[i];
    }
    if constexpr (ROLE == 1) {
    sfor<MAXSLOT>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      if (ti[s] >= 0) {
        if (jq.Ldense) {
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int rr = 16 * ti[s] + lr + 4 * v, cc = 16 * tj[s] + lc;
            if (rr < nb && cc < nb) {
              const int ro = jq.flip ? n - 1 - rr : rr, rz = jq.flip ? n - 1 - cc : cc;  // (flip: no border row, nb == n)
              jq.Ldense[(size_t)ro * jq.ldo + cc] = (cc <= rr) ? tile[s][v] : 0.0;
              if (ti[s] != tj[s]) jq.Ldense[(size_t)rz * jq.ldo + rr] = 0.0;
            }
          }
        }
        if (jq.Lpack && 16 * ti[s] < n && 16 * tj[s] < n) c2_pack_tile(jq.Lpack, tile[s], ti[s], tj[s], n, lr, lc);
      }
    });
    }
    if (jq.piv_out)
      for (int i = tid; i < n; i += C2_WAVES * 64) jq.piv_out[i] = S.pivs[i];
    if (jq.Dinv_out) {
      // inverses of the diagonal blocks in the layout k_fwdsub reads (identity at / behind the border row): with Lpack this is
      // everything the covariance product behind an update needs from chol(T)
      const int ntn = (n + 15) >> 4;
      if constexpr (ROLE == 0)
        if (!jq.y_out) c2_invert_diag_blocks(S, 0, ntn, wave, lr, lc);  // (the back substitution has inverted them already)
      __syncthreads();
      if constexpr (ROLE == 0) {
        for (int e = tid; e < ntn * 256; e += C2_EW * 64) {
          const int k = e >> 8, i = (e >> 4) & 15, c = e & 15;
          const int gr = 16 * k + i, gc = 16 * k + c;
          jq.Dinv_out[e] = (gr < n && gc < n) ? S.Dsave[k * C2_TSZ + i * C2_TS + c] : (i == c ? 1.0 : 0.0);
        }
      }
    }
    return;
  }

  if (jq.mode == 2) {
    // ---- range part of the plane's residual:  pr = |Lr^-1 bn|^2, rank deficiency of the normalised Gram ----
    // Rank: with the regularisation eps on the unit diagonal a deficient direction shows up as a pivot of eps x (1 .. 1e5) - the
    // factor is |v|^2 / v_c^2 for the null vector v completed at column c - while the pivots of the well-determined directions
    // stay above 1e-3 (measured on config-3 sized planes, tools/plane_gate_agreement.py): eps = 1e-12 and a threshold of 1e-5
    // leave two decades on either side.  (The columns arrive with the clones last, so the deficiency of the gauge freedom
    // completes in the trailing pivots.)
    if (wave == 0) {
      double pr = 0.0;
      for (int i = lane; i < n; i += 64) pr = fma(S.zbuf[i], S.zbuf[i], pr);
      pr = wave_sum(pr);
      // rank deficiency = pivots below tol_strict + pivots below tol_loose behind the first of those, counted by the whole wave
      // (one lane walking the ~200 pivots in LDS was 20 K cycles at the end of this workgroup - the gate of the other one waited)
      int first = 0x7fffffff, n_strict = 0;
      for (int i = lane; i < pq.n_involved; i += 64) {
        if (S.pivs[i] < pq.tol_strict) {
          ++n_strict;
          first = i < first ? i : first;
        }
      }
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) {
        n_strict += __shfl_xor(n_strict, m);
        const int o = __shfl_xor(first, m);
        first = o < first ? o : first;
      }
      int n_loose = 0;
      for (int i = lane; i < pq.n_involved; i += 64) {
        const double pv = S.pivs[i];
        if (i > first && !(pv < pq.tol_strict) && pv < pq.tol_loose) ++n_loose;
      }
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) n_loose += __shfl_xor(n_loose, m);
      if (lane == 0) {
        const int ndeg = n_strict + n_loose;
        pq.scal[1] = pr;
        pq.scal[2] = (double)ndeg;
        __threadfence();
        __hip_atomic_store(pq.range_done, pq.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    return;
  }

  // ---- mode 1: plane update.  zz = |Lt^-1 c|^2 = b . dx ----
#define M1_STAMP(i) C2_STAMP(nt + 1, i)
  M1_STAMP(0);
  if (wave == 0) {
    double zz = 0.0;
    for (int i = lane; i < n; i += 64) zz = fma(S.zbuf[i], S.zbuf[i], zz);
    zz = wave_sum(zz);
    if (lane == 0 && part == 0) {
      // part A: its share of |z|^2 and the verdict on its pivots go to part B, which decides; then wait for the decision
      pq.xzz[0] = zz;
      pq.xzz[1] = (double)bad;
      __hip_atomic_store(pq.xsync, pq.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      int spins = 0;
      unsigned v = 0u;
      bool timed_out = false;
      while (((v = __hip_atomic_load(pq.xsync + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) >> 1) != (pq.seq & 0x7fffffffu)) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1 << 19)) {
          timed_out = true;
          break;
        }
      }
      if (timed_out && jq.flag) atomicOr(jq.flag, 2);
      sh_zz = zz;
      sh_ok = (!timed_out && (v & 1u)) ? 1 : 0;
    } else if (lane == 0) {
      // the other workgroup publishes pr and the rank; both take the same time, this wait is short.  Bounded (~0.3 s): HIP gives
      // no forward-progress guarantee between two workgroups of a grid (CU masking down to one CU, a scheduler that starts block
      // 1 only when block 0 retires) - then the plane is rejected and the call fails with OVP_E_TIMEOUT instead of hanging.
      int spins = 0;
      bool timed_out = false;
      if (part == 1) {  // the other half of |z|^2
        while (__hip_atomic_load(pq.xsync, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != pq.seq) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > (1 << 19)) {
            timed_out = true;
            break;
          }
        }
        zz += pq.xzz[0];
        bad |= (int)pq.xzz[1];
        spins = 0;
      }
      while (__hip_atomic_load(pq.range_done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != pq.seq) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1 << 18)) {
          timed_out = true;
          break;
        }
      }
      if (timed_out && jq.flag) atomicOr(jq.flag, 2);
      const double rr = pq.scal[0], pr = pq.scal[1], ndeg = pq.scal[2];
      // The reference keeps rows_u rows of the Givens-compressed system of which (rows_u - rank) carry no Jacobian: each is a
      // combination of the rows below it weighted by the ROUNDING NOISE those rows hold in a deficient column.  The m identical
      // constraint rows of a feature leave m - 1 rows that are zero to the last bit (no noise, no weight), so the retained rows
      // sample the rows_live = sum(2m - 2) directions that carry residual energy, not all 3m - 3 of them
      // (tools/plane_gate_study.py: energy per retained junk row 0.98 against 0.64 per stacked row on config-3 planes).
      const double rank = (double)pq.n_involved - ndeg;
      const double noise_rows = fmax((double)pq.rows_u - rank, 0.0);
      const double denom = (double)pq.rows_live - rank;
      const double frac = denom > 0.5 ? fmin(noise_rows / denom, 1.0) : 1.0;
      const double chi2 = (pr - zz) + pq.noise_scale * frac * fmax(rr - pr, 0.0);
      // any failure upstream (chol(P) of the loop's start, an earlier plane's factorization) rejects this and every later plane
      // BEFORE anything is committed: with a failed L0 the tables and the covariance would otherwise drift apart
      const int upstream = jq.flag ? __hip_atomic_load(jq.flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
      const bool fact_ok = (bad == 0) && !timed_out && upstream == 0;
      const bool ok = fact_ok && (pq.force == 0 ? false : (pq.force == 1 ? true : (chi2 <= pq.thr)));
      pq.res_out[0] = chi2;
      pq.res_out[1] = ok ? 1.0 : 0.0;
      pq.res_out[2] = ndeg;
      pq.res_out[3] = pr;
      sh_zz = zz;
      sh_ok = ok ? 1 : 0;
      if (part == 1 && !ok)  // part A is waiting for the decision
        __hip_atomic_store(pq.xsync + 1, (pq.seq & 0x7fffffffu) << 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
  if (!sh_ok) return;  // rejected: nothing changes (update/UpdaterMSCKF.cpp:613-631)
  M1_STAMP(1);

  const int ntn = (n + 15) >> 4;           // tile rows of the n x n part
  const int h_bs = part >= 0 ? (jq.split_h < ntn ? jq.split_h : ntn) : 0;
  if (part == 0)  // y blocks of part B's columns (published before its decision)
    for (int i = 16 * h_bs + tid; i < 16 * ntn; i += C2_WAVES * 64) S.ybuf[i] = pq.xy[i];
  chol2_backsolve<NS, ROLE>(S, n, ntn, tile, ti, tj, nullptr, part == 0 ? h_bs : ntn, part == 1 ? h_bs : 0);
  M1_STAMP(2);
  if (part == 1) {
    // ---- part B ends here: its y blocks and the decision go to part A, its share of the factor to the buffers behind the loop ----
    for (int i = 16 * h_bs + tid; i < 16 * ntn; i += C2_WAVES * 64) pq.xy[i] = S.ybuf[i];
    __syncthreads();
    if (tid == 0) {
      const bool fine = S.cnt[6] == 0;  // no hand-over of the back substitution timed out
      if (!fine) {
        if (jq.flag) atomicOr(jq.flag, 2);
        pq.res_out[1] = 0.0;
      }
      __hip_atomic_store(pq.xsync + 1, ((pq.seq & 0x7fffffffu) << 1) | (fine ? 1u : 0u), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (pq.cond && pq.emit && S.cnt[6] == 0) {
      if constexpr (ROLE == 1) {
        sfor<MAXSLOT>([&](auto sc) {
          constexpr int s = decltype(sc)::value;
          if (ti[s] >= 0 && 16 * ti[s] < n && 16 * tj[s] < n) c2_pack_tile(pq.Lpack, tile[s], ti[s], tj[s], n, lr, lc, pq.n_full);
        });
      } else {
        for (int e = 256 * h_bs + tid; e < ntn * 256; e += C2_EW * 64) {
          const int k = e >> 8, i = (e >> 4) & 15, c = e & 15;
          const int gr = 16 * k + i, gc = 16 * k + c;
          pq.Dinv[e] = (gr < n && gc < n) ? S.Dsave[k * C2_TSZ + i * C2_TS + c] : (i == c ? 1.0 : 0.0);
        }
      }
    }
    return;
  }
  // dx = L0 y  (L0 dense lower triangular, row-major).  Two threads per row, each streaming half of the row's non-zeros with
  // 16-byte loads that are all in flight together (a wave-per-row loop serialised one L2 round trip per row: 40 us).
  const int nfull = pq.n_full > n ? pq.n_full : n;  // rows of L0 (the factorization may have run on the leading n columns only)
  double* dxs = nfull > n ? S.PB : S.zbuf;  // z is no longer needed; the panel buffers (free behind the back substitution) when
                                            // the correction is longer than the factorized dimension
  __syncthreads();
  if (S.cnt[6]) {  // a hand-over of the back substitution timed out: nothing is committed, the call fails
    if (tid == 0) {
      if (jq.flag) atomicOr(jq.flag, 2);
      pq.res_out[1] = 0.0;
    }
    return;
  }
  {
    // Round 5: wave w owns rows w, w + 12, ...; a row's non-zeros are read as 1 KB pieces (64 lanes x 16 bytes: eight cache lines per
    // vector-memory instruction - the two-threads-per-row form this replaces touched 64 lines per instruction and was bound by the
    // access rate of the CU's vector L1: 15.2 K cycles for 230 KB), every piece of a batch of rows in flight together, the lane's own
    // pairs of y in registers, one transposed reduction per batch.  Batches by row length (rows < 120: one piece, < 252: two, < 288:
    // three) so that the register arrays are static.
    const int wv = tid >> 6, ln = tid & 63;
    dbl2_t yp[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int c0 = 2 * (ln + 64 * k);
      yp[k][0] = c0 < n ? S.ybuf[c0] : 0.0;
      yp[k][1] = c0 + 1 < n ? S.ybuf[c0 + 1] : 0.0;
    }
    auto batch = [&](auto jlo_c, auto cnt_c, auto k_c) {
      constexpr int JLO = decltype(jlo_c)::value, CNT = decltype(cnt_c)::value, K = decltype(k_c)::value;
      constexpr int NP = CNT <= 2 ? 2 : (CNT <= 4 ? 4 : (CNT <= 8 ? 8 : 16));
      if (12 * JLO >= nfull) return;  // (uniform: no row of this batch exists)
      dbl2_t v[CNT][K];
#pragma unroll
      for (int j = 0; j < CNT; ++j) {
        const int row = wv + 12 * (JLO + j);
        const int lastc = row < n ? row : n - 1;  // last column of the row's non-zeros that meets y
        const int npair = (lastc + 2) >> 1;       // 16-byte pairs covering columns 0..lastc (ld is even, rows are 16-byte aligned)
        const dbl2_t* lrow = reinterpret_cast<const dbl2_t*>(pq.L0 + (size_t)row * pq.ld0);
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const int q = ln + 64 * k;
          v[j][k] = (row < nfull && q < npair) ? lrow[q] : dbl2_t{0.0, 0.0};
        }
      }
      double part[NP];
#pragma unroll
      for (int j = 0; j < NP; ++j) part[j] = 0.0;
#pragma unroll
      for (int j = 0; j < CNT; ++j) {
        const int row = wv + 12 * (JLO + j);
        const int lastc = row < n ? row : n - 1;
        double sa = 0.0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const int c0 = 2 * (ln + 64 * k);
          sa = fma(v[j][k][0], yp[k][0], sa);
          sa = fma(c0 + 1 <= lastc ? v[j][k][1] : 0.0, yp[k][1], sa);
        }
        part[j] = sa;
      }
      const double tot = wave_transpose_reduce<NP>(part);  // lane L: the sum of part[L / (64 / NP)] over the wave
      const int jo = ln / (64 / NP);
      const int row = wv + 12 * (JLO + jo);
      if ((ln % (64 / NP)) == 0 && jo < CNT && row < nfull) dxs[row] = tot;
    };
    batch(std::integral_constant<int, 0>{}, std::integral_constant<int, 8>{}, std::integral_constant<int, 1>{});
    batch(std::integral_constant<int, 8>{}, std::integral_constant<int, 2>{}, std::integral_constant<int, 1>{});
    batch(std::integral_constant<int, 10>{}, std::integral_constant<int, 4>{}, std::integral_constant<int, 2>{});
    batch(std::integral_constant<int, 14>{}, std::integral_constant<int, 4>{}, std::integral_constant<int, 2>{});
    batch(std::integral_constant<int, 18>{}, std::integral_constant<int, 3>{}, std::integral_constant<int, 2>{});
    batch(std::integral_constant<int, 21>{}, std::integral_constant<int, 3>{}, std::integral_constant<int, 3>{});
  }
  __syncthreads();
  M1_STAMP(3);
  // ---- commit (ext Type::update on the device tables, update/UpdaterMSCKF.cpp:646-648) ----
  for (int i = tid; i < nfull; i += C2_WAVES * 64) {
    pq.dx_out[i] = dxs[i];
    pq.dx_last[i] = dxs[i];
  }
  if (tid == 0) *pq.cur ^= 1;  // the accumulated T of this plane becomes the current one
  for (int i = tid; i < pq.n_feat_local; i += C2_WAVES * 64) pq.feat_used[pq.feat_list[i]] = 1;
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
  if (tid < pq.n_clones) {
    const int id = pq.clone_id[tid];
    rot_update(pq.clone_R + 9 * tid, dxs + id);
    for (int k = 0; k < 3; ++k) pq.clone_p[3 * tid + k] += dxs[id + 3 + k];
  } else if (tid == pq.n_clones) {
    if (pq.calib_id >= 0) {
      rot_update(pq.cal, dxs + pq.calib_id);
      for (int k = 0; k < 3; ++k) pq.cal[9 + k] += dxs[pq.calib_id + 3 + k];
    }
    if (pq.intr_id >= 0)
      for (int k = 0; k < 8; ++k) pq.cal[12 + k] += dxs[pq.intr_id + k];
  } else if (tid > pq.n_clones && tid <= pq.n_clones + pq.n_planes) {
    const int pl = tid - pq.n_clones - 1;
    if (pq.plane_sid[pl] >= 0)
      for (int k = 0; k < 3; ++k) pq.cp[3 * pl + k] += dxs[pq.plane_sid[pl] + k];
  }
  for (int q = tid; q < pq.n_slam; q += C2_WAVES * 64)
    for (int k = 0; k < 3; ++k) pq.slam_p[3 * q + k] += dxs[pq.slam_id[q] + k];
  M1_STAMP(4);
  // ---- the factor for the covariance product behind the loop ----
  if (pq.cond) {
    if (tid == 0) pq.cond[1] = pq.seq_plane;
    if (pq.emit) {
      if constexpr (ROLE == 1) {
        sfor<MAXSLOT>([&](auto sc) {
          constexpr int s = decltype(sc)::value;
          if (ti[s] >= 0 && 16 * ti[s] < n && 16 * tj[s] < n) c2_pack_tile(pq.Lpack, tile[s], ti[s], tj[s], n, lr, lc, pq.n_full);
        });
      } else {
        // inverses of the diagonal blocks (the back substitution left them in S.Dsave), identity at / behind the border row
        // (a split factorization: part B wrote the blocks of its columns)
        for (int e = tid; e < (part == 0 ? h_bs : ntn) * 256; e += C2_EW * 64) {
          const int k = e >> 8, i = (e >> 4) & 15, c = e & 15;
          const int gr = 16 * k + i, gc = 16 * k + c;
          pq.Dinv[e] = (gr < n && gc < n) ? S.Dsave[k * C2_TSZ + i * C2_TS + c] : (i == c ? 1.0 : 0.0);
        }
        if (tid == 0) pq.cond[0] = pq.seq_plane;
      }
    }
  }
}

template <int MAXSLOT, bool SPLIT>
__global__ __launch_bounds__(C2_WAVES * 64) void k_chol2(const Chol2Job J0, const Chol2Job J1, const PlaneSolve ps) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  __shared__ Chol2Shared sh;
  const Chol2Job& J = blockIdx.x == 1 ? J1 : J0;  // block 2 = part B of a split plane update (same job as block 0)
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  // both instantiations execute the same sequence of workgroup barriers
  const int joff = blockIdx.x == 1 ? (int)sizeof(Chol2Job) : 0;  // where J sits in the kernel-argument segment (chol2_body)
  if (wave < C2_EW) chol2_body<MAXSLOT, 0, SPLIT>(J, ps, lds, sh, joff);
  else chol2_body<MAXSLOT, 1, SPLIT>(J, ps, lds, sh, joff);
}

// out[0] = max_i A_ii (one workgroup): the scale the drop threshold of a semi-definite factorization refers to
__global__ __launch_bounds__(256) void k_max_diag(const double* __restrict__ A, int n, int ld, double* __restrict__ out) {
  __shared__ double red[256];
  double m = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) m = fmax(m, A[(size_t)i * ld + i]);
  red[threadIdx.x] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0];
}

}  // namespace ovp

extern "C" {

hipError_t ovp_launch_max_diag(const double* A, int n, int ld, double* out, hipStream_t stream) {
  hipLaunchKernelGGL(ovp::k_max_diag, dim3(1), dim3(256), 0, stream, A, n, ld, out);
  return hipGetLastError();
}

int ovp_chol2_max_n(void) { return 16 * 18 - 1; }  // bordered dimension n + 1 <= 288 (18 tile rows)

// chol(T) of an EKF update as k_tilechol delivers it (tile-packed factor + inverted diagonal blocks for k_fwdsub), on the
// second-generation kernel: fused elimination, role hand-over through LDS counters (74 against 94 us at N = 240)
hipError_t ovp_launch_chol2_packed(const double* A, double* Dinv, double* Lpack, int n, int ld, int* flag, int add_identity,
                                   const int* cond, hipStream_t stream) {
  ovp::Chol2Job j;
  memset(&j, 0, sizeof(j));
  j.A = A;
  j.n = n;
  j.ld = ld;
  j.add_identity = add_identity;
  j.mode = 0;
  j.flag = flag;
  j.Lpack = Lpack;
  j.Dinv_out = Dinv;
  j.skip_cond = cond;
  return ovp_launch_chol2(&j, nullptr, nullptr, stream);
}

// one workgroup (j1 == nullptr), two (plane loop: update part and range part side by side) or three (the update part split over
// two workgroups, j0->split_h > 0: block 0 = tile columns < split_h, block 2 = the rest)
int ovp_chol2_stamps_compiled(void) { return C2_STAMPS_ON; }

hipError_t ovp_launch_chol2(const ovp::Chol2Job* j0, const ovp::Chol2Job* j1, const ovp::PlaneSolve* ps, hipStream_t stream) {
  using namespace ovp;
  auto ntof = [](const Chol2Job* j) { return ((j->brow ? j->n + 1 : j->n) + 15) / 16; };
  int nt = ntof(j0);
  int tiles = nt * (nt + 1) / 2;
  if (j0->split_h > 0) {
    const int h = j0->split_h;
    if (!j1 || !ps || j0->mode != 1 || h >= nt || !j0->xbuf || !j0->xflag || !ps->xzz || !ps->xy || !ps->xsync) return hipErrorInvalidValue;
    const int ta = h * nt - (h * (h - 1)) / 2;
    tiles = ta > tiles - ta ? ta : tiles - ta;
  }
  if (j1) {
    const int nt1 = ntof(j1);
    if (nt1 * (nt1 + 1) / 2 > tiles) tiles = nt1 * (nt1 + 1) / 2;
    if (nt1 > nt) nt = nt1;
  }
  const int slots = (tiles + C2_TW - 1) / C2_TW;
  const size_t shmem = (size_t)chol2_lds_doubles(nt) * sizeof(double);
  Chol2Job dummy = *j0;
  PlaneSolve psd = PlaneSolve();
  const Chol2Job& b = j1 ? *j1 : dummy;
  const PlaneSolve& p = ps ? *ps : psd;
  const dim3 grid(j0->split_h > 0 ? 3 : (j1 ? 2 : 1)), block(C2_WAVES * 64);
  static unsigned long long attr_mask = 0;  // per device (ovp_kernels.h)
  if (ovp_lds_attr_needed(&attr_mask)) {
    (void)hipFuncSetAttribute((const void*)k_chol2<10, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)k_chol2<15, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)k_chol2<17, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)k_chol2<20, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)k_chol2<22, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)k_chol2<13, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)k_chol2<16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)k_chol2<22, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipGetLastError();  // (a kernel with static LDS refuses the full 160 KB: harmless, a real shortage fails the launch itself)
    ovp_lds_attr_done(&attr_mask);
  }
  if (j0->split_h > 0) {  // the split kernels: the slot counts of BASELINE config 3 (forced, tests) and config 4, and a catch-all
    if (slots <= 13)
      hipLaunchKernelGGL((k_chol2<13, true>), grid, block, shmem, stream, *j0, b, p);
    else if (slots <= 16)
      hipLaunchKernelGGL((k_chol2<16, true>), grid, block, shmem, stream, *j0, b, p);
    else if (slots <= 22)
      hipLaunchKernelGGL((k_chol2<22, true>), grid, block, shmem, stream, *j0, b, p);
    else
      return hipErrorInvalidValue;
    return hipGetLastError();
  }
  if (slots <= 10)
    hipLaunchKernelGGL((k_chol2<10, false>), grid, block, shmem, stream, *j0, b, p);
  else if (slots <= 15)
    hipLaunchKernelGGL((k_chol2<15, false>), grid, block, shmem, stream, *j0, b, p);
  else if (slots <= 17)
    hipLaunchKernelGGL((k_chol2<17, false>), grid, block, shmem, stream, *j0, b, p);
  else if (slots <= 20)
    hipLaunchKernelGGL((k_chol2<20, false>), grid, block, shmem, stream, *j0, b, p);
  else if (slots <= 22)  // n + 1 up to 288: the tile registers no longer fit 168 VGPRs, part of them lives in scratch
    hipLaunchKernelGGL((k_chol2<22, false>), grid, block, shmem, stream, *j0, b, p);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}
}
