// Device-side helpers shared by the gfx950 kernels of libovplane_hip.so (wave64 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define OVP_WAVE 64
#define OVP_REC 21  // per-row record: clone block (6) | calibration block (14) | residual (1)

namespace ovp {

// ---- cross-lane primitives (wave64) -----------------------------------------------------------
__device__ __forceinline__ double shfl_xor_f64(double v, int mask) {
  // ds_bpermute based exchange of a 64-bit value
  int lo = __double2loint(v), hi = __double2hiint(v);
  const int src = ((int)(threadIdx.x & 63) ^ mask) << 2;
  lo = __builtin_amdgcn_ds_bpermute(src, lo);
  hi = __builtin_amdgcn_ds_bpermute(src, hi);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double readlane_f64(double v, int lane) {
  // lane must be wave-uniform
  int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// value of lane (lane ^ S), S = 8, 4, 2 or 1, through DPP (plain VALU; ds_bpermute goes through the LDS crossbar, ~130 cycles):
// row_ror:8 is the exchange of the two halves of a 16-lane row, xor 4 is row_shr:4 into the upper quads of each half and
// row_shl:4 into the lower ones (bank masks), xor 2 / xor 1 are quad permutations
template <int S>
__device__ __forceinline__ int xor_lane_b32(int x) {
  static_assert(S == 8 || S == 4 || S == 2 || S == 1, "DPP exchange inside a 16-lane row");
  if constexpr (S == 8) return __builtin_amdgcn_update_dpp(0, x, 0x128, 0xF, 0xF, true);
  else if constexpr (S == 4) {
    const int t = __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xA, false);
    return __builtin_amdgcn_update_dpp(t, x, 0x104, 0xF, 0x5, false);
  } else if constexpr (S == 2) return __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true);
  else return __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true);
}
template <int S>
__device__ __forceinline__ double xor_lane_f64(double v) {
  return __hiloint2double(xor_lane_b32<S>(__double2hiint(v)), xor_lane_b32<S>(__double2loint(v)));
}

// Transpose-reduce: every lane passes v[0..N) (N a power of two <= 64); on return element 0 of lane L holds
// sum over all 64 lanes of v[k(L)], where k(L) = top log2(N) bits of the lane id (bit 5 most significant).
// The remaining (64/N)-lane groups are then combined by a plain butterfly, so every lane of a group holds the
// full sum.  Costs N-1 + log2(64/N) exchanges instead of 6*N.
template <int H, int S>
struct TransposeReduceStep {
  template <int N>
  static __device__ __forceinline__ void run(double (&v)[N], int lane) {
    if constexpr (S == 32 || S == 16) {
      // the gfx950 half / row swaps do the exchange AND the selection: with A = v[k], B = v[k + H], v_permlane32_swap leaves
      // A' = [A(0..31) | B(0..31)], B' = [A(32..63) | B(32..63)], so A' + B' is A's sum in the lower half and B's in the upper one
      // (v_permlane16_swap: the same per pair of 16-lane rows).  Plain VALU - the ds_bpermute pair + four selects per exchange
      // went through the LDS crossbar (~130 cycles a round trip); same operands per sum, so the results are bit-identical.
#pragma unroll
      for (int k = 0; k < H; ++k) {
        const unsigned alo = (unsigned)__double2loint(v[k]), ahi = (unsigned)__double2hiint(v[k]);
        const unsigned blo = (unsigned)__double2loint(v[k + H]), bhi = (unsigned)__double2hiint(v[k + H]);
        if constexpr (S == 32) {
          auto lo = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
          auto hi = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
          v[k] = __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
        } else {
          auto lo = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
          auto hi = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
          v[k] = __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
        }
      }
    } else {
      const bool up = (lane & S) != 0;
#pragma unroll
      for (int k = 0; k < H; ++k) {
        const double keep = up ? v[k + H] : v[k];
        const double send = up ? v[k] : v[k + H];
        v[k] = keep + xor_lane_f64<S>(send);
      }
    }
    if constexpr (H > 1) TransposeReduceStep<H / 2, S / 2>::run(v, lane);
  }
};
template <int N>
__device__ __forceinline__ double wave_transpose_reduce(double (&v)[N]) {
  static_assert(N >= 2 && N <= 64 && (N & (N - 1)) == 0, "N must be a power of two");
  const int lane = threadIdx.x & 63;
  TransposeReduceStep<N / 2, 32>::run(v, lane);
  double r = v[0];
#pragma unroll
  for (int s = 32 / N; s >= 1; s >>= 1) r += shfl_xor_f64(r, s);
  return r;
}
// lane that holds (after wave_transpose_reduce<N>) the sum of element k: any lane whose top bits equal k
template <int N>
__device__ __forceinline__ int reduce_owner_lane(int k) {
  return k * (64 / N);
}

// broadcast of lane N of every 16-lane DPP row to all lanes of that row (row_share:N, gfx90a+): the value stays in
// VGPRs, so unlike v_readlane there is no VALU -> SGPR -> VALU round trip on the dependent chain.
template <int N>
__device__ __forceinline__ double row_share_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x150 + N, 0xF, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x150 + N, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}

// exchange with the neighbouring lane (lane ^ 1) through DPP quad_perm [1,0,3,2]
__device__ __forceinline__ double swap_pair_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  // old = 0 with bound_ctrl: every lane is written, so no copy of the source into the destination is needed first
  lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) v += shfl_xor_f64(v, s);
  return v;
}

// bitwise OR of a 64-bit mask over the wave (every lane gets the result)
__device__ __forceinline__ unsigned long long wave_or_u64(unsigned long long v) {
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) {
    const int src = ((int)(threadIdx.x & 63) ^ s) << 2;
    const unsigned lo = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)(unsigned)(v & 0xffffffffull));
    const unsigned hi = (unsigned)__builtin_amdgcn_ds_bpermute(src, (int)(unsigned)(v >> 32));
    v |= ((unsigned long long)hi << 32) | lo;
  }
  return v;
}

// sum over the four 16-lane rows of the wave (lanes l, l + 16, l + 32, l + 48), every lane gets the result: the gfx950 row /
// half swaps are plain VALU operations (a ds_bpermute round trip through the LDS crossbar costs ~130 cycles each)
__device__ __forceinline__ double rows_sum_f64(double v) {
  unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  auto l16 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  auto h16 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  v = __hiloint2double((int)h16[0], (int)l16[0]) + __hiloint2double((int)h16[1], (int)l16[1]);
  lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  auto l32 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  auto h32 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double((int)h32[0], (int)l32[0]) + __hiloint2double((int)h32[1], (int)l32[1]);
}

// index into a packed lower-triangular matrix (row i >= col j)
__device__ __forceinline__ int tri(int i, int j) { return (i * (i + 1)) / 2 + j; }

}  // namespace ovp
