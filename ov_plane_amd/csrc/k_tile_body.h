// Device code of the register-resident tile Cholesky (see k_tile.hip for the description); shared by k_tilechol and by the
// fused feature + chol(P) kernel of k_feat.hip.
#pragma once
#include "ovp_dev.h"
#include "ovp_kernels.h"
#include <utility>
#include <cstdlib>

namespace ovp {

typedef double double4_t __attribute__((ext_vector_type(4)));

template <int... Is, class F>
__device__ __forceinline__ void sfor_impl(std::integer_sequence<int, Is...>, F&& f) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
  sfor_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

// opaque use + redefinition of a value: keeps the optimizer from sinking / hoisting the computation across this point
__device__ __forceinline__ void pin_vgpr(double& v) { asm volatile("" : "+v"(v)); }

__device__ __forceinline__ double rsqrt_nr2(double x) {
  double y = __builtin_amdgcn_rsq(x);
  const double h = 0.5 * x;
  y = y * fma(-h * y, y, 1.5);
  y = y * fma(-h * y, y, 1.5);
  return y;
}

static constexpr int TS = 17;        // LDS row pitch of a 16x16 tile (doubles)
static constexpr int TSZ = 16 * TS;  // doubles per LDS tile
static constexpr int TC_WAVES = 8;       // 1 factor wave + 7 tile waves
static constexpr int TC_TILE_WAVES = 7;

// multiply-accumulate of two LDS tiles in "row, k" form:  acc += sign * X[row][:] . Y[col][:]
__device__ __forceinline__ double4_t mfma_xyT(const double* X, const double* Y, double4_t acc, double sign, int lc,
                                              int lr) {
  // all eight operands first, then two accumulation chains: a dependent f64 MFMA costs ~200 cycles, an independent one
  // ~110, and every tile update sits in its own scalar-branch block, so nothing else could fill the gaps
  double a[4], b[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    a[s] = sign * X[lc * TS + lr + 4 * s];
    b[s] = Y[lc * TS + lr + 4 * s];
  }
  double4_t acc1 = {0.0, 0.0, 0.0, 0.0};
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b[0], acc, 0, 0, 0);
  acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], b[1], acc1, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[2], b[2], acc, 0, 0, 0);
  acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[3], b[3], acc1, 0, 0, 0);
  return acc + acc1;
}

// 16x16 diagonal block, one wave.
//  (1) Cholesky with lane r <-> row r (the four 16-lane DPP rows run identical copies): right-looking, the column just
//      finished is broadcast with v_readlane (measured faster here than DPP row_share or an LDS round trip), the next
//      pivot is taken first so that its rsq/Newton chain overlaps the rest of the step.
//  (2) X = L^-1 with lane c <-> COLUMN c of X: sixteen independent forward substitutions, the entries of L are
//      wave-uniform operands (16-byte LDS broadcasts from a column-major copy written in (1)), so the inverse costs
//      136 lane-local FMAs and no cross-lane traffic (carrying the identity through (1) took 2 v_readlane per FMA).
// Outputs: Dbuf = L_kk (zero above the diagonal), Wbuf = X^T (the MFMA B-operand of the panel solve), dinv_out = X.
__device__ __forceinline__ void diag_factor(double* Dbuf, double* Wbuf, double* Sbuf, double* dinv_out, int lane,
                                            int* flag) {
  const int row = lane & 15;
  double d[16];
  sfor<16>([&](auto cc) {
    constexpr int c = decltype(cc)::value;
    d[c] = Dbuf[row * TS + c];
  });
  bool bad = false;
  double piv = readlane_f64(d[0], 0);
  double inv = rsqrt_nr2(piv);
  sfor<16>([&](auto cc) {
    constexpr int c = decltype(cc)::value;
    bad = bad || !(piv > 0.0);
    const double inv_c = inv;
    const double l = d[c] * inv_c;  // column c of L for rows >= c
    d[c] = l;
    // column-major copy for (2) with 1 / L_cc on the diagonal; rows < c are never read
    if (lane < 16) Sbuf[c * 16 + row] = (row == c) ? inv_c : l;
    if constexpr (c + 1 < 16) {
      const double l1 = readlane_f64(l, c + 1);
      d[c + 1] = fma(-l, l1, d[c + 1]);
      piv = readlane_f64(d[c + 1], c + 1);
      inv = rsqrt_nr2(piv);
    }
    if constexpr (c + 2 < 16) {
      // L[j][c], j = c+2..15, come back from the column-major copy as 16-byte LDS broadcasts (operands in VGPRs: the
      // v_readlane form needs 2 SGPRs per element and spills the scalar file)
      constexpr int e0 = (c + 2) & ~1;
      typedef double dbl2 __attribute__((ext_vector_type(2)));
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      const dbl2* col = reinterpret_cast<const dbl2*>(Sbuf + c * 16 + e0);
      dbl2 lv[(16 - e0) / 2];
#pragma unroll
      for (int q = 0; q < (16 - e0) / 2; ++q) lv[q] = col[q];
      sfor<14 - c>([&](auto jc) {
        constexpr int j = c + 2 + decltype(jc)::value;
        d[j] = fma(-l, lv[(j - e0) >> 1][(j - e0) & 1], d[j]);
      });
    }
    __builtin_amdgcn_sched_barrier(0);  // one column at a time
  });
  if (bad && lane == 0) *flag = 1;
  if (lane < 16) {
    sfor<16>([&](auto cc) {
      constexpr int c = decltype(cc)::value;
      Dbuf[row * TS + c] = (c <= row) ? d[c] : 0.0;
    });
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // Sbuf written by lanes 0..15 of this wave, read by all below
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  // (2) column `row` of X:  x_m = e_m / L_mm after eliminating rows 0..m-1;  x_i -= L_im x_m  for i > m
  double x[16];
  int rowv = row;
  asm volatile("" : "+v"(rowv));  // opaque: keeps the 16 identity selects out of the (loop-invariant) spill area
  sfor<16>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    x[i] = (i == rowv) ? 1.0 : 0.0;
  });
  sfor<16>([&](auto mc) {
    constexpr int m = decltype(mc)::value;
    constexpr int e0 = m & ~1;  // aligned pairs starting at the even index <= m (the diagonal slot holds 1 / L_mm)
    typedef double dbl2 __attribute__((ext_vector_type(2)));
    const dbl2* col = reinterpret_cast<const dbl2*>(Sbuf + m * 16 + e0);
    dbl2 lv[(16 - e0) / 2];
#pragma unroll
    for (int q = 0; q < (16 - e0) / 2; ++q) lv[q] = col[q];
    const double xm = x[m] * lv[(m - e0) >> 1][(m - e0) & 1];
    x[m] = xm;
    sfor<15 - m>([&](auto ic) {
      constexpr int i = m + 1 + decltype(ic)::value;
      x[i] = fma(-lv[(i - e0) >> 1][(i - e0) & 1], xm, x[i]);
    });
    // one column of broadcast reads at a time: the optimizer otherwise hoists all 64 of them, or sinks the FMAs into the
    // `lane < 16` block below, and spills either way
    sfor<16 - m>([&](auto ic) {
      constexpr int i = m + decltype(ic)::value;
      pin_vgpr(x[i]);
    });
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  });
  if (lane < 16) {
    sfor<16>([&](auto rc) {
      constexpr int r = decltype(rc)::value;
      Wbuf[row * TS + r] = x[r];                       // W = X^T : W[c][r] = X[r][c], lane = c
      if (dinv_out) dinv_out[r * 16 + row] = x[r];     // X row-major
    });
  }
}

// Wave 0 is the factor wave (owns no tiles, so the 16x16 factorization does not compete with the tile registers);
// waves 1..TC_TILE_WAVES hold the tiles.  Both roles execute exactly two workgroup barriers per step.
// `lds` needs tilechol_lds_doubles(nt) doubles; the caller is a workgroup of TC_WAVES waves (k_tilechol, or block 0 of the
// fused feature kernel in k_feat.hip).
__host__ __device__ constexpr int tilechol_lds_doubles(int nt) { return (2 + nt + TC_TILE_WAVES) * TSZ + 256 + 2; }

template <int MAXSLOT>
__device__ __forceinline__ void tilechol_body(const double* __restrict__ A, double* __restrict__ L,
                                              double* __restrict__ Dinv, double* __restrict__ Lpack, int n, int ld,
                                              int* __restrict__ flag, int add_identity, int dbg_skip, double* lds,
                                              const int flip = 0, double* __restrict__ boost = nullptr, const int boost_n = 0,
                                              const double boost_rel = 0.0) {
  // flip: factorize the matrix in REVERSED index order and store the dense factor with its rows reversed back,
  // Lr[n - 1 - r][c] = chol(J A J)[r][c] (J = exchange matrix): Lr Lr^T = A, and column j of Lr is zero below row n - 1 - j.
  // A point update whose information matrix lives on the TRAILING columns [s0, n) of the state (clones and calibration behind
  // the IMU block) then has  T = I + Lr^T A Lr = blockdiag(T_lead, I)  with a leading block of n - s0 columns - in the state's
  // own order, without a permutation of P (ovp_api_point.hip: ekf_from_gram).
  const int nt = (n + 15) >> 4;
  const int ntiles = nt * (nt + 1) / 2;
  double* Dbuf = lds;
  double* Wbuf = Dbuf + TSZ;
  double* PB = Wbuf + TSZ;
  double* SW = PB + nt * TSZ;
  double* Sbuf = SW + TC_TILE_WAVES * TSZ;        // 256 doubles: column-major L_kk for the inverse (factor wave only)
  int* rdy = reinterpret_cast<int*>(Sbuf + 256);  // diagonal tile k is in Dbuf once *rdy >= k + 1
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane >> 4, lc = lane & 15;
  if (dbg_skip & 32) return;  // diagnostics: launch overhead only
  if (tid == 0) *rdy = 0;
  __syncthreads();

  // Look-ahead: the diagonal tile of step k+1 is brought up to date FIRST in the trailing phase of step k and handed to
  // the factor wave through an LDS flag, so its 16x16 factorization (the longest serial piece of a step) runs while the
  // tile waves finish the rest of the trailing update.  Two workgroup barriers per step:
  //   B2(k): L_kk / W published (factor wave -> tile waves)      B3(k): panel published (tile waves -> tile waves)
  if (wave == 0) {
    for (int k = 0; k < nt; ++k) {
      while (__hip_atomic_load(rdy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < k + 1) __builtin_amdgcn_s_sleep(1);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      if (!(dbg_skip & 1)) diag_factor(Dbuf, Wbuf, Sbuf, Dinv ? Dinv + (size_t)k * 256 : nullptr, lane, flag);
      __syncthreads();  // B2
      __syncthreads();  // B3
    }
  } else {
    const int tw = wave - 1;
    double* sw = SW + tw * TSZ;
    double4_t tile[MAXSLOT];
    int ti[MAXSLOT], tj[MAXSLOT];
    // ---- load: tile index idx = slot*TILE_WAVES + tw, column-major over the lower tile triangle ----
    // (i, j) of consecutive slots are found incrementally on the scalar unit (they depend on the wave index only; as
    // SGPRs they also make the per-step "is this a panel / trailing / diagonal tile" tests scalar branches).  Loads
    // are branch-free - out-of-range elements read a clamped address and are replaced by the identity padding - so all
    // of a wave's tiles are in flight together.
    const int nfull = n >> 4;  // tiles with index < nfull need no range checks
    int jj = 0, cstart = 0;    // current tile column and the index of its first tile
    sfor<MAXSLOT>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      const int idx = s * TC_TILE_WAVES + tw;
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
      double4_t t = {0.0, 0.0, 0.0, 0.0};
      if (i >= 0 && (dbg_skip & 8)) {
        if (i == j) t[0] = (lc == lr) ? 4.0 : 0.0, t[1] = (lc == lr + 4) ? 4.0 : 0.0, t[2] = (lc == lr + 8) ? 4.0 : 0.0,
                    t[3] = (lc == lr + 12) ? 4.0 : 0.0;
      } else if (i >= 0) {
        const int c = 16 * j + lc;
        const int cc = c < n ? c : n - 1;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int r = 16 * i + lr + 4 * v;
          const int rc = r < n ? r : n - 1;
          double x = flip ? A[(size_t)(n - 1 - rc) * ld + (n - 1 - cc)] : A[(size_t)rc * ld + cc];
          if (i >= nfull) x = (r < n && c < n) ? x : 0.0;  // i >= j: a partial tile is always in the last tile row
          if (r == c) x = (r < n) ? (add_identity ? x + 1.0 : x) : 1.0;  // identity padding keeps the matrix SPD
          if (flip && r == c && r < n && r >= n - boost_n) {  // CholJob::boost: state column n - 1 - r < boost_n
            const double add = x * boost_rel;
            boost[n - 1 - r] = add;
            x += add;
          }
          t[v] = x;
        }
      }
      tile[s] = t;
    });

    // hands the (up to date) diagonal tile kk to the factor wave
    auto publish_diag = [&](const double4_t& t, int kk) {
#pragma unroll
      for (int v = 0; v < 4; ++v) Dbuf[(lr + 4 * v) * TS + lc] = t[v];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) __hip_atomic_store(rdy, kk + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    sfor<MAXSLOT>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      if (ti[s] == 0 && tj[s] == 0) publish_diag(tile[s], 0);
    });

    // Store of a finished tile: the tile-packed copy (column-major inside the tile = coalesced MFMA A-operand reads in
    // k_fwdsub) and / or the dense L (lower; the mirrored tile of the strict upper triangle is zeroed).  What stores cost here
    // is their NUMBER - ~30 cycles of the CU's one vector-memory pipe each, 16 us of an 86 us factorization at N = 210 when
    // every tile took twelve 8-byte stores, wherever in the kernel they were issued - so the tile is turned around in the
    // wave's LDS scratch (free between the trailing update and the next panel solve) and leaves as 16-byte stores, and a
    // caller that only needs the packed copy passes L = nullptr.  (ld, lc, lr: opaque per-step copies, otherwise the address
    // arithmetic of all 15 slots is hoisted out of the step loop and spills.)
    typedef double dbl2s __attribute__((ext_vector_type(2)));
    auto store_tile = [&](auto sc, int ld, int lc, int lr) {
      constexpr int s = decltype(sc)::value;
      if (ti[s] >= 0 && !(dbg_skip & 16)) {
#pragma unroll
        for (int v = 0; v < 4; ++v) sw[(lr + 4 * v) * TS + lc] = tile[s][v];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const int l2 = 2 * (16 * lr + lc);  // two consecutive elements per lane and store
        if (Lpack) {
          double* pk = Lpack + (size_t)(s * TC_TILE_WAVES + tw) * (ld - ld + 256);
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int e = l2 + 128 * h;  // packed offset = column * 16 + row
            const int col = e >> 4, row = e & 15;
            *reinterpret_cast<dbl2s*>(pk + e) = dbl2s{sw[row * TS + col], sw[(row + 1) * TS + col]};
          }
        }
        if (!L) {
        } else if (ti[s] < nfull) {  // interior tile (j <= i < nfull): no range checks
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int e = l2 + 128 * h;  // row-major inside the tile
            const int row = e >> 4, col = e & 15;
            const int ro = 16 * ti[s] + row, rz = 16 * tj[s] + row;
            *reinterpret_cast<dbl2s*>(L + (size_t)(flip ? n - 1 - ro : ro) * ld + 16 * tj[s] + col) =
                dbl2s{sw[row * TS + col], sw[row * TS + col + 1]};
            if (ti[s] != tj[s]) *reinterpret_cast<dbl2s*>(L + (size_t)(flip ? n - 1 - rz : rz) * ld + 16 * ti[s] + col) = dbl2s{0.0, 0.0};
          }
        } else {
          const int c = 16 * tj[s] + lc;
          const int c2 = 16 * ti[s] + lc;
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int r = 16 * ti[s] + lr + 4 * v;
            if (r < n && c < n) L[(size_t)(flip ? n - 1 - r : r) * ld + c] = tile[s][v];
          }
          if (ti[s] != tj[s]) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
              const int r2 = 16 * tj[s] + lr + 4 * v;
              if (r2 < n && c2 < n) L[(size_t)(flip ? n - 1 - r2 : r2) * ld + c2] = 0.0;
            }
          }
        }
        __builtin_amdgcn_wave_barrier();  // sw is reused by the next tile
      }
    };

    for (int k = 0; k < nt; ++k) {
      __syncthreads();  // B2: L_kk in Dbuf, W = L_kk^-T in Wbuf
      // (c) diagonal owner reloads L_kk; panel tiles <- tile * W, published to PB
      sfor<MAXSLOT>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        if (tj[s] == k) {
          if (ti[s] == k) {
#pragma unroll
            for (int v = 0; v < 4; ++v) tile[s][v] = Dbuf[(lr + 4 * v) * TS + lc];
          } else if (ti[s] > k && !(dbg_skip & 2)) {
#pragma unroll
            for (int v = 0; v < 4; ++v) sw[(lr + 4 * v) * TS + lc] = tile[s][v];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            double4_t acc = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
            double pa[4], pw[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              pa[q] = sw[lc * TS + lr + 4 * q];
              pw[q] = Wbuf[(lr + 4 * q) * TS + lc];
            }
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[0], pw[0], acc, 0, 0, 0);  // two chains, see mfma_xyT
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[1], pw[1], acc1, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[2], pw[2], acc, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[3], pw[3], acc1, 0, 0, 0);
            acc = acc + acc1;
            tile[s] = acc;
            double* pb = PB + ti[s] * TSZ;
#pragma unroll
            for (int v = 0; v < 4; ++v) pb[(lr + 4 * v) * TS + lc] = acc[v];
            __builtin_amdgcn_wave_barrier();
          }
        }
      });
      __syncthreads();  // B3: panel k in PB (Dbuf / Wbuf of step k are dead from here on)
      // (d) trailing update, next diagonal tile first
      sfor<MAXSLOT>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        if (ti[s] == k + 1 && tj[s] == k + 1) {
          if (!(dbg_skip & 4)) tile[s] = mfma_xyT(PB + ti[s] * TSZ, PB + tj[s] * TSZ, tile[s], -1.0, lc, lr);
          publish_diag(tile[s], k + 1);
        }
      });
      sfor<MAXSLOT>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        if (tj[s] > k && !(ti[s] == k + 1 && tj[s] == k + 1) && !(dbg_skip & 4))
          tile[s] = mfma_xyT(PB + ti[s] * TSZ, PB + tj[s] * TSZ, tile[s], -1.0, lc, lr);
      });
      // column k is final: its tiles go out while the following steps run
      int ld_k = __builtin_amdgcn_readfirstlane(ld), lc_k = lc, lr_k = lr;
      asm volatile("" : "+s"(ld_k));
      asm volatile("" : "+v"(lc_k), "+v"(lr_k));
      sfor<MAXSLOT>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        if (tj[s] == k) store_tile(sc, ld_k, lc_k, lr_k);
      });
      // PB is rewritten only after B2 of the next step, which every wave reaches after its trailing update
    }
    // (tiles were stored as their columns became final, see store_tile)
  }
}

}  // namespace ovp
