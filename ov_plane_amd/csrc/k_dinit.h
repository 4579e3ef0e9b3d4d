// Parameter block of the delayed-initialisation loop kernel (k_dinit.hip; update/UpdaterSLAM.cpp:204-364).
#pragma once
#include "ovp_kernels.h"

#define OVP_DINIT_DYN_LDS (138 * 1024)  // dynamic LDS of k_dinit_rows (160 KB minus its static tables)

namespace ovp {

struct DinitParams {
  FeatParams fp;          // the candidates as a feature batch (uv, clone_idx, n_meas, p_FinG), pose tables, 1 / sigma_pix, do_fej, calmask
  int cand;               // candidate index in the batch, -1 = commit only (behind the last candidate)
  int m_obs;              // its observations (n_meas[cand]; known to the host, spares the kernel a dependent load)
  int n, n_max;           // covariance dimension in front of this candidate; capacity (layout of the result blocks)
  double* P;              // resident covariance (fp.ldp)
  double* clone_R;        // writable pose tables (the commit of the previous candidate)
  double* clone_p;
  double* cal;
  const double* prev_res; // result block of the previous candidate [chi2 | accepted | negdiag | - | dx (n) ...], nullptr = none
  const int* ids;         // [cols] state columns of the candidate's H_x: clone blocks in observation order, estimated calibration
  int idv[208];           // the same list by value (scalar reads in the kernel; 6 * OVP_MAX_MEAS + 14 = 206 entries at most)
  // outputs for k_init_m / k_init_core / k_init_update
  double* Ht;             // [cols][rows]
  double* Hinv;           // [9] H_L^-1 row-major
  double* Rk;             // [9] R_init = I
  double* resid;          // [rows - 3] residual of the update rows
  double* res;            // this candidate's result block: res[4 + n_max .. +3) = H_L^-1 res_init
};

}  // namespace ovp

extern "C" {
size_t ovp_dinit_rows_lds(int m_obs, int ncal);
hipError_t ovp_launch_dinit_rows(const ovp::DinitParams* dp, size_t lds, hipStream_t stream);
}
