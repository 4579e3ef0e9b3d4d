// Parameter block of the SLAM-landmark update kernel (k_slam.hip; update/UpdaterSLAM.cpp:424-673).
#pragma once
#include "ovp_kernels.h"

namespace ovp {

struct SlamParams {
  FeatParams fp;             // the landmarks' measurements as a feature batch (uv, clone_idx, n_meas, p_FinG = Landmark::get_xyz(false)),
                             // the pose tables, the resident covariance, 1 / sigma_pix, chi2_multipler, the quantile table
  const double* p_fej;       // [L][3] Landmark::get_xyz(true)
  const int* lm_id;          // [L] Type::id() of the landmark
  const int* plane_sid;      // [L] Type::id() of the in-state plane the landmark lies on, -1 = none (nullptr = no planes at all)
  const double* cp;          // [L][3] its closest point, value / first estimate (State::_features_PLANE)
  const double* cp_fej;
  double white_c;            // 1 / sigma_constraint
  // landmarks whose dense block the host built (anchored / inverse-depth representations): rows > 0 marks them
  const int* pre_rows;       // [L] or nullptr
  const int* pre_cols;       // [L]
  const int* pre_off;        // [L] offset (doubles) of the block in pre_H: [rows x cols] column-major, then res [rows]
  const int* pre_ids_off;    // [L] offset of its column ids in pre_ids
  const double* pre_H;
  const int* pre_ids;
  // the stacked system
  const int* row0;           // [L] first stacked row of the landmark
  const int* gpos;           // [n] state column -> row of Ht (position in the call's column list)
  double* Ht;                // [gcols = columns of the call][m_total]; every row of the stack is written by its landmark's block
  int m_total, gcols;
  double* res_out;           // [m_total]
  double* Mall;              // [n][m_total] = P[:, columns] H^T for the S-form update (k_init.hip), nullptr = not needed
  // scratch / geometry
  double* Hscr;              // [L][rows_max * cols_max] when the block does not fit LDS (h_in_lds = 0)
  int rows_max, cols_max, h_in_lds;
  // per-landmark results
  double* chi2;              // [L] statistic of the stage that decided
  unsigned char* status;     // [L] 0 = rejected, 1 = accepted, 2 = accepted without its plane (fallback)
};

}  // namespace ovp

extern "C" {
size_t ovp_slam_gate_lds(int rows_max, int cols_max, int with_h);
hipError_t ovp_launch_slam_gate(const ovp::SlamParams* sp, int n_landmarks, size_t lds, hipStream_t stream);
}
