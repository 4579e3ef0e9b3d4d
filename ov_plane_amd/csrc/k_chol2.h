// Parameter blocks of k_chol2 (k_chol2.hip): second-generation tile Cholesky with a bordered right-hand side, and the
// plane-loop epilogues that run inside it.
#pragma once
#include <hip/hip_runtime.h>

namespace ovp {

struct Chol2Job {
  const double* A;      // n x n, lower triangle read (row-major, leading dimension ld)
  int n, ld;
  int add_identity;     // factorize A + I
  int mode;             // 0 = factor only, 1 = plane update (gate / back substitution / commit), 2 = plane range part
  const int* sel;       // optional: A += ((*sel) ^ sel_xor) * sel_stride  (double-buffered input chosen on the device)
  int sel_xor;
  size_t sel_stride;
  const double* brow;   // optional border row [n]: z = L^-1 brow^T is produced along the way
  int* flag;            // set to 1 on a non-positive pivot
  const double* floor_scale;  // optional device scalar: the effective floor is piv_floor * (*floor_scale)
  double piv_floor;     // > 0: columns whose pivot falls below it are dropped (rank-deficient semi-definite systems: mode 2), never flagged
  // mode 0 outputs (any may be null)
  double* Lpack;        // tile-packed factor of the n x n part (the layout k_fwdsub reads)
  double* Ldense;       // dense factor of the bordered matrix, (n + 1) x ldo
  int ldo;
  double* z_out;        // [n]
  double* y_out;        // [n] L^-T z
  double* piv_out;      // [n] pivots before the square root
  double* Dinv_out;     // inverted diagonal blocks of the n x n part, [ceil(n/16)][16][16] (what k_fwdsub reads beside Lpack)
  const int* skip_cond; // optional {have, want}: return at once when the factor this launch would produce is already there
  long long* stamps;    // optional [nt + 1][8] cycle stamps of wave 0 (diagnostics)
  int dbg;              // timing experiments, -DOVP_C2_STAMPS builds only: 1 = skip the fused elimination, 2 = skip the trailing MFMAs
  // mode 1 on two workgroups (k_chol2.hip, chol2_factor): tile columns < split_h on block 0, the rest on block 2
  int split_h;          // 0 = one workgroup
  double* xbuf;         // [split_h][nt][256] exported panel tiles (rows >= split_h), row-major
  unsigned* xflag;      // [nt] <- xseq when the panel of that step is exported
  unsigned xseq;
  // mode 0, no border row: factorize the matrix in REVERSED index order and write the dense factor with its rows reversed back
  // (Ldense = Lr, Lr Lr^T = A, columns in the reversed order - what CholJob::flip of the first-generation body produces, see
  // ovp_kernels.h); the diagonal of the first boost_n state columns is read as (1 + boost_rel) x its value, the added amounts left
  // in boost[0 .. boost_n)
  int flip;
  double* boost;
  int boost_n;
  double boost_rel;
};

// per-plane arguments of the plane loop's solve (modes 1 and 2)
struct PlaneSolve {
  // gate (update/UpdaterMSCKF.cpp:606-631)
  double* scal;             // [0] = rr (all projected residual rows), [1] = pr, [2] = rank deficiency  (device)
  unsigned* range_done;     // sequence word: the range workgroup stores `seq` when scal[1..2] are valid
  unsigned seq;
  double thr;
  int rows_live, rows_u, n_involved;  // rows_live: residual directions that can carry energy (2m - 2 per feature, see the gate)
  int force;                // ovp_plane_batch::force_decision (0 / 1), anything else = the gate decides
  double noise_scale;       // weight of the expected energy of the rounding-decided rows (OVP_PLANE_NOISE_KAPPA; a study may override it)
  double tol_strict, tol_loose;
  double* res_out;          // [4]: chi2, accept, rank deficiency, pr
  // split factorization: [0] part A's share of |z|^2, [1] its pivot verdict; y blocks of part B's columns; sequence words
  // ([0] <- seq: xzz valid, [1] <- 2 seq + accept: decision taken, xy valid)
  double* xzz;
  double* xy;
  unsigned* xsync;
  // solution and commit
  const double* L0;         // factor of the covariance at the start of the loop, dense lower triangular
  int ld0;
  int n_full;               // rows of L0 / of the correction (0 = the factorized dimension): the factorization may run on the LEADING
                            // n < n_full columns when everything behind them has never been involved (T = blockdiag(T_lead, I));
                            // dx = L0[:, 0:n] y, the tile-packed factor is laid out for n_full with the identity behind n
  double* dx_out;           // [n] correction of this plane (for the host)
  double* dx_last;          // [n] scratch copy on the device
  int* cur;                 // index of the current accumulated-T buffer, toggled on accept
  const int* feat_list;     // features of this plane ...
  int n_feat_local;
  unsigned char* feat_used; // ... marked as consumed on accept
  double *clone_R, *clone_p;
  const int* clone_id;
  int n_clones;
  double* cal;
  int calib_id, intr_id;
  double* cp;
  const int* plane_sid;
  int n_planes;
  int n_slam;
  const int* slam_id;
  double* slam_p;
  // the factor of the accepted T for the covariance product behind the loop (k_fwdsub's inputs): cond[1] = seq_plane on every
  // accept; with emit != 0 the plane also leaves Lpack / Dinv behind and sets cond[0] = seq_plane - the k_tilechol behind the loop
  // (ovp_launch_tilechol_unless) then finds its work done if this plane stays the last accepted one
  int emit, seq_plane;
  double *Lpack, *Dinv;
  int* cond;
};

}  // namespace ovp

extern "C" {
int ovp_chol2_max_n(void);
int ovp_chol2_stamps_compiled(void);  // 1 in a build under -DOVP_C2_STAMPS (tools/build_c2_stamps.sh): Chol2Job::stamps is written
hipError_t ovp_launch_max_diag(const double* A, int n, int ld, double* out, hipStream_t stream);
hipError_t ovp_launch_chol2_packed(const double* A, double* Dinv, double* Lpack, int n, int ld, int* flag, int add_identity,
                                   const int* cond, hipStream_t stream);
hipError_t ovp_launch_chol2(const ovp::Chol2Job* j0, const ovp::Chol2Job* j1, const ovp::PlaneSolve* ps, hipStream_t stream);
}
