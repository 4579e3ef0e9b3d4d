// Parameter blocks and launchers of the second-generation plane loop (k_plane2.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "ovp_kernels.h"

#include "ovplane_hip.h"
#define PA_MAXQ OVP_PLANE_MAX_SLAM  // SLAM landmarks on one out-of-state plane handled per update

namespace ovp {

struct PlaneAsm {
  const double* gramS;  // [n_clones][n_chunks][OVP_GRAM_ELEMS] per-clone Grams of the sparse rows (k_gram_pair)
  int n_clones, n_chunks;
  const double* part;   // [n_split][ntile][256] G^T G partials
  int n_split, ntile;
  const ColMap* colmap;
  int n;                // state dimension
  int plane_sid;        // Type::id() of the plane, -1 when it is not in the state
  int in_state;
  const double* cst;    // [nf][10] constraint-row moments (k_plane_feat)
  int nf;
  // SLAM landmarks on planes outside the state
  int n_slam, plane1;
  const int* slam_plane;
  const int* slam_id;
  const double* slam_p;
  const double* slam_p_fej;
  const double* cp;      // this plane's closest point (value / first estimate)
  const double* cp_fej;
  double white_c;
  int do_fej;
  // outputs
  double* Ab;            // (n + 1) x lda
  int lda;
  const int* perm;       // [n] position among the involved columns, -1 = not involved
  double* An;            // n_inv x ldn normalised, regularised Gram in `perm` order
  int ldn;
  double* bn;            // [n_inv]
  double eps;
  double* scal;          // [0] <- total projected residual energy
};

}  // namespace ovp

extern "C" {
hipError_t ovp_launch_plane_assemble2(const ovp::PlaneAsm* a, hipStream_t stream);
hipError_t ovp_launch_plane_dT(int n, const double* L0, int ld, const double* W, const double* b, double* Tbuf, size_t tstride,
                               const int* cur, double* crow, hipStream_t stream);
hipError_t ovp_launch_plane_sub_accum(const double* res, const double* Ab, double* Asum, const double* dx, double* u, int ns, int ld,
                                      hipStream_t stream);
// clears up to eight device regions (sizes in bytes, multiples of 4) in ONE launch; fill_regions: pattern[i] 0 = zeros, 1 = 16 x 16
// identity blocks, 2 = tile-packed lower triangle (ntn tile rows) with identity diagonal tiles (doubles)
hipError_t ovp_launch_zero_regions(void* const* ptr, const size_t* bytes, int count, hipStream_t stream);
hipError_t ovp_launch_fill_regions(void* const* ptr, const size_t* bytes, const int* pattern, int count, int ntn, hipStream_t stream);
hipError_t ovp_launch_select_copy(double* dst, const double* buf, size_t stride, const int* cur, int n, int ld, int sym,
                                  hipStream_t stream);
}
