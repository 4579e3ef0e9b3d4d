// DPP row-broadcast forms of the f64 FMA / move (gfx90a+), as inline asm: the 16-column elimination of k_chol2.hip and of K1
// (k_feat.hip) is built from them.
#pragma once
#include <hip/hip_runtime.h>

namespace ovp {

// d[j] += bcast_j(nl) * l   (lane j of the own 16-lane row supplies nl).  A DPP read of a VGPR needs two wait states behind the
// VALU write of that register, and inline asm is invisible to the hazard recognizer: the plain form may only be used where the
// wait states are guaranteed otherwise (they came with the write, or a data dependency orders this statement behind one that
// carries them) - source order guarantees nothing, asm statements without `volatile` may be emitted in any order.
#define C2_FMAC_DPP(acc, src_dpp, mul, J)                                                                      \
  asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #J " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src_dpp), "v"(mul))

#define C2_FMAC_DPP_NOP(acc, src_dpp, mul, J)                                                                       \
  asm("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:" #J " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src_dpp), "v"(mul))

// same, preceded by the two wait states
template <int J>
__device__ __forceinline__ void fmac_bcast_nop(double& acc, const double& src_dpp, const double& mul) {
  if constexpr (J == 0) C2_FMAC_DPP_NOP(acc, src_dpp, mul, 0);
  else if constexpr (J == 1) C2_FMAC_DPP_NOP(acc, src_dpp, mul, 1);
  else if constexpr (J == 2) C2_FMAC_DPP_NOP(acc, src_dpp, mul, 2);
  else if constexpr (J == 3) C2_FMAC_DPP_NOP(acc, src_dpp, mul, 3);
  else if constexpr (J == 4) C2_FMAC_DPP_NOP(acc, src_dpp, mul, 4);
  else if constexpr (J == 5) C2_FMAC_DPP_NOP(acc, src_dpp, mul, 5);
  else if constexpr (J == 6) C2_FMAC_DPP_NOP(acc, src_dpp, mul, 6);
  else if constexpr (J == 7) C2_FMAC_DPP_NOP(acc, src_dpp, mul, 7);
  else if constexpr (J == 8) C2_FMAC_DPP_NOP(acc, src_dpp, mul, 8);
  else if constexpr (J == 9) C2_FMAC_DPP_NOP(acc, src_dpp, mul, 9);
  else if constexpr (J == 10) C2_FMAC_DPP_NOP(acc, src_dpp, mul, 10);
  else if constexpr (J == 11) C2_FMAC_DPP_NOP(acc, src_dpp, mul, 11);
  else if constexpr (J == 12) C2_FMAC_DPP_NOP(acc, src_dpp, mul, 12);
  else if constexpr (J == 13) C2_FMAC_DPP_NOP(acc, src_dpp, mul, 13);
  else if constexpr (J == 14) C2_FMAC_DPP_NOP(acc, src_dpp, mul, 14);
  else C2_FMAC_DPP_NOP(acc, src_dpp, mul, 15);
}

template <int J>
__device__ __forceinline__ void fmac_bcast(double& acc, const double& src_dpp, const double& mul) {
  if constexpr (J == 0) C2_FMAC_DPP(acc, src_dpp, mul, 0);
  else if constexpr (J == 1) C2_FMAC_DPP(acc, src_dpp, mul, 1);
  else if constexpr (J == 2) C2_FMAC_DPP(acc, src_dpp, mul, 2);
  else if constexpr (J == 3) C2_FMAC_DPP(acc, src_dpp, mul, 3);
  else if constexpr (J == 4) C2_FMAC_DPP(acc, src_dpp, mul, 4);
  else if constexpr (J == 5) C2_FMAC_DPP(acc, src_dpp, mul, 5);
  else if constexpr (J == 6) C2_FMAC_DPP(acc, src_dpp, mul, 6);
  else if constexpr (J == 7) C2_FMAC_DPP(acc, src_dpp, mul, 7);
  else if constexpr (J == 8) C2_FMAC_DPP(acc, src_dpp, mul, 8);
  else if constexpr (J == 9) C2_FMAC_DPP(acc, src_dpp, mul, 9);
  else if constexpr (J == 10) C2_FMAC_DPP(acc, src_dpp, mul, 10);
  else if constexpr (J == 11) C2_FMAC_DPP(acc, src_dpp, mul, 11);
  else if constexpr (J == 12) C2_FMAC_DPP(acc, src_dpp, mul, 12);
  else if constexpr (J == 13) C2_FMAC_DPP(acc, src_dpp, mul, 13);
  else if constexpr (J == 14) C2_FMAC_DPP(acc, src_dpp, mul, 14);
  else C2_FMAC_DPP(acc, src_dpp, mul, 15);
}

#define C2_MOV_DPP(dst, src, J) \
  asm("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:" #J " row_mask:0xf bank_mask:0xf" : "=v"(dst) : "v"(src))

template <int J>
__device__ __forceinline__ double bcast_row(const double& src) {
  double dst;
  if constexpr (J == 0) C2_MOV_DPP(dst, src, 0);
  else if constexpr (J == 1) C2_MOV_DPP(dst, src, 1);
  else if constexpr (J == 2) C2_MOV_DPP(dst, src, 2);
  else if constexpr (J == 3) C2_MOV_DPP(dst, src, 3);
  else if constexpr (J == 4) C2_MOV_DPP(dst, src, 4);
  else if constexpr (J == 5) C2_MOV_DPP(dst, src, 5);
  else if constexpr (J == 6) C2_MOV_DPP(dst, src, 6);
  else if constexpr (J == 7) C2_MOV_DPP(dst, src, 7);
  else if constexpr (J == 8) C2_MOV_DPP(dst, src, 8);
  else if constexpr (J == 9) C2_MOV_DPP(dst, src, 9);
  else if constexpr (J == 10) C2_MOV_DPP(dst, src, 10);
  else if constexpr (J == 11) C2_MOV_DPP(dst, src, 11);
  else if constexpr (J == 12) C2_MOV_DPP(dst, src, 12);
  else if constexpr (J == 13) C2_MOV_DPP(dst, src, 13);
  else if constexpr (J == 14) C2_MOV_DPP(dst, src, 14);
  else C2_MOV_DPP(dst, src, 15);
  return dst;
}


}  // namespace ovp
