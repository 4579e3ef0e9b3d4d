// Internal header of the C-ABI shim (libovplane_hip.so): the context, the launch prototypes of the kernels' translation units and
// the helpers the entry-point files share.  Not installed; the boundary is include/ovplane_hip.h.
//   ovp_api_ctx.hip    context, covariance residency and bookkeeping (upload / marginal / propagate / clone / marginalize /
//                      initialize), pose tables, feature batch, diagnostics
//   ovp_api_point.hip  point update: K1 -> information pair -> EKF update (ovp_msckf_update and its staged form), ovp_ekf_update
//   ovp_api_rccl.hip   feature-sharded update over RCCL
//   ovp_api_plane.hip  plane loop (ovp_msckf_plane_update) and plane initialisation (ovp_plane_init)
//   ovp_api_slam.hip   SLAM landmarks (ovp_slam_update, ovp_slam_delayed_init), triangulation
#pragma once
#include "ovplane_hip.h"

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <vector>

#include "ovp_kernels.h"
#include "k_chol2.h"
#include "k_plane2.h"
#include "k_slam.h"
#include "k_dinit.h"

extern "C" int ovp_dbg_tilechol_skip;
extern "C" {
hipError_t ovp_launch_scatter_gram_add(const double* Acc, const double* bcc, int cols, const int* col_ids, double* Ab, int lda, int n,
                                       hipStream_t stream);
hipError_t ovp_launch_scatter_gram(const double* Acc, const double* bcc, int cols, const int* col_ids, double* Ab,
                                   int lda, int n, hipStream_t stream);
hipError_t ovp_launch_gather_marginal(const double* P, int ldp, const int* cols, int m, double* out, hipStream_t stream);
hipError_t ovp_launch_gather_block(const double* P, int ldp, const int* ids, int m, double* out, int ldo, hipStream_t stream);
hipError_t ovp_launch_gather_block_boost(const double* P, int ldp, const int* ids, int m, double* out, int ldo, int from, double rel,
                                         double* boost, hipStream_t stream);
hipError_t ovp_launch_gather_block_unless(const double* P, int ldp, const int* ids, int m, double* out, int ldo, const int* cancel,
                                          hipStream_t stream);
hipError_t ovp_launch_unit_diag(const double* P, int n, int ld, double* C, double* dvec, hipStream_t stream);
hipError_t ovp_launch_unpermute_pair(const double* Pperm, const double* V, int ld, const int* ids, int n, double* Pout, double* Lout,
                                     int ldo, const int* cancel, const double* boost, hipStream_t stream);
hipError_t ovp_launch_factor_from_V(const double* V, int ld, const int* ids, int n, double* out, int ldo, hipStream_t stream);
hipError_t ovp_launch_scale_rows(double* L, int n, int ld, const double* dvec, hipStream_t stream);
hipError_t ovp_launch_gather_cols(const double* P, int ldp, const int* ids, int n, int m, double* G, int ldg, hipStream_t stream);
hipError_t ovp_launch_mat_sub(const double* A, const double* B, double* C, int rows, int cols, int ld, hipStream_t stream);
hipError_t ovp_launch_sub_sym(double* P, const double* D, int n, int ld, hipStream_t stream);
hipError_t ovp_launch_sub_sym_unless(double* P, const double* D, int n, int ld, const int* cancel, hipStream_t stream);
hipError_t ovp_launch_cov_clone(double* P, int ldp, int n_old, int src, int sz, double jitter, hipStream_t stream);
hipError_t ovp_launch_cov_marginalize(const double* src, double* dst, int ld, int n_old, int id, int sz,
                                      hipStream_t stream);
hipError_t ovp_launch_propagate(double* P, int ldp, int n, int start, int phi, const int* oldcol, int nold,
                                const double* Phi, const double* Q, double* CPT, double* PCP, int* negdiag,
                                hipStream_t stream);
hipError_t ovp_launch_propagate_publish(double* P, int ldp, int n, int start, int phi, const int* oldcol, int nold, const double* Phi,
                                        const double* Q, double* CPT, double* PCP, int* negdiag, unsigned* host_dev, unsigned seq,
                                        hipStream_t stream);
hipError_t ovp_launch_augment_dt(double* P, int ldp, int n, int pose, int dt, const double* d, hipStream_t stream);
hipError_t ovp_launch_init_invertible(double* P, int ldp, int n, const int* cols, int ncols, const double* HR, int k,
                                      double* Ma, const double* Hinv, const double* Rk, hipStream_t stream);
hipError_t ovp_launch_tilechol_unless(const double* A, double* L, double* Dinv, double* Lpack, int n, int ld, int* flag,
                                      int add_identity, const int* cond, hipStream_t stream);
hipError_t ovp_launch_tilechol(const double* A, double* L, double* Dinv, double* Lpack, int n, int ld, int* flag, int add_identity,
                               hipStream_t stream);
hipError_t ovp_launch_fwdsub_lead(const double* Ltp, const double* Dinv, const double* Lmat, double* V, int n, int ld, int dense,
                                  int n_lead, hipStream_t stream);
hipError_t ovp_launch_fwdsub(const double* Lt, const double* Dinv, const double* Lmat, double* V, int n, int ld,
                             int dense, hipStream_t stream);
hipError_t ovp_launch_gemm4(int transA, int transB, int M, int N, int K, const double* A, int lda, const double* B,
                            int ldb, double* C, int ldc, int add_identity, int symmetric, hipStream_t stream);
hipError_t ovp_launch_dx_rows_boost(double* P, int n, int ldp, const double* b, double* dx, int* negdiag, unsigned* ticket,
                                    void* res_block, void* host_block, int pub_words, void* seq_host, unsigned seq,
                                    const double* boost, int boost_n, const int* cancel, hipStream_t stream);
hipError_t ovp_launch_dx_rows(const double* P, int n, int ldp, const double* b, double* dx, int* negdiag,
                              unsigned* ticket, void* res_block, void* host_block, int pub_words, void* seq_host,
                              unsigned seq,
                              hipStream_t stream);
hipError_t ovp_launch_reduce_gram(const double* gramS, int n_clones, int n_chunks, double* gramR, hipStream_t stream);
hipError_t ovp_launch_plane_feat(const ovp::FeatParams* p, const ovp::PlaneParams* pp, int n_local, hipStream_t stream);
hipError_t ovp_launch_reduce_cst(const double* cst, int nf, double* out, hipStream_t stream);
hipError_t ovp_launch_assemble_ext(const double* gramR, int n_clones, const double* part, int n_split,
                                   const ovp::ColMap* colmap, int n, int plane_sid, const double* cstsum, double* E,
                                   int lde, hipStream_t stream);
hipError_t ovp_launch_plane_reduce_to_state(const double* E, int lde, int n, int in_state, double* Ab, int lda,
                                            const double* rr_in, double* scal, hipStream_t stream);
hipError_t ovp_launch_normalize_reg(const double* Ab, int lda, int n, double eps, double* An, double* bn,
                                    hipStream_t stream);
hipError_t ovp_launch_range_energy(const double* Lr, const double* Dinv, const double* bn, int n, int ld, double tol,
                                   double* scal, hipStream_t stream);
hipError_t ovp_launch_dx_from_factor(const double* V, int n, int ld, const double* b, double* dx, double* scal,
                                     hipStream_t stream);
hipError_t ovp_launch_init_m(const double* P, int ldp, int n, const int* ids, int cols, const double* Ht, int m, double* Mall,
                             hipStream_t stream);
hipError_t ovp_launch_init_core(double* P, int ldp, int n, const int* ids, int cols, const double* Ht, int k, int rup, double* Mall,
                                const double* Hinv, const double* Rk, const double* resid, double r_iso, double thr, double* Linv,
                                double* y, double* res, hipStream_t stream);
hipError_t ovp_launch_init_update(const double* Psrc, double* Pdst, int ldp, int n2, const double* Mall, int m, int k, int rup,
                                  const double* Linv, const double* y, double* res, double* dx, hipStream_t stream);
size_t ovp_init_core_lds(int k, int rup, int cols);
size_t ovp_init_max_lds();
int ovp_init_max_rows();
hipError_t ovp_launch_gemm4c(int transA, int transB, int M, int N, int K, const double* A, int lda, const double* B, int ldb,
                             double* C, int ldc, int add_identity, int symmetric, const int* cancel, hipStream_t stream);
hipError_t ovp_launch_plane_gate(const double* scal, const int* flags, double thr, int rows_live, int rows_u,
                                 int n_involved, int force, double* res_out, hipStream_t stream);
hipError_t ovp_launch_plane_init_augment(const double* E, int lde, int ns, const int* ids, int n, double* P, int ldp, const double* dx, double* out,
                                         hipStream_t stream);
hipError_t ovp_launch_plane_slam_rows(double* E, int lde, int n, int plane1, int n_slam, const int* slam_plane, const int* slam_id,
                                      const double* slam_p, const double* slam_p_fej, const double* cp, const double* cp_fej,
                                      double white_c, int do_fej, double* cstsum, hipStream_t stream);
hipError_t ovp_launch_plane_commit_slam(const double* res, const double* dx, int n_slam, const int* slam_id, double* slam_p,
                                        hipStream_t stream);
hipError_t ovp_launch_plane_commit(const double* res, const double* V, double* M, int n, int ld, const double* dx,
                                   double* dx_out, double* clone_R, double* clone_p, const int* clone_id, int n_clones,
                                   double* cal, int calib_id, int intr_id, double* cp, const int* plane_sid,
                                   int n_planes, hipStream_t stream);
}


static const int OVP_TILECHOL_NMAX = 288;  // register-resident factorization limit (22 tiles per wave)

#define HIPCHK(x)                               \
  do {                                          \
    hipError_t _e = (x);                        \
    if (_e != hipSuccess) return (int)_e;       \
  } while (0)

static inline int round_up(int v, int m) { return ((v + m - 1) / m) * m; }

// ------------------------------------------------------------------------------------------------
struct ovp_ctx {
  int device = 0;
  hipStream_t stream = nullptr, stream2 = nullptr;
  bool own_stream = false;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  hipEvent_t ev_t[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  int n_max = 0, c_max = 0, f_max = 0;
  int n = 0, ld = 0;  // current covariance size, leading dimension of every n x n buffer
  double *P = nullptr, *P_tmp = nullptr;
  // state tables
  double *clone_R = nullptr, *clone_p = nullptr, *clone_R_fej = nullptr, *clone_p_fej = nullptr;
  int* clone_id = nullptr;
  double* cal = nullptr;  // [20] camera extrinsics / intrinsics values
  ovp::ColMap* colmap = nullptr;
  double* chi2_table = nullptr;
  ovp::FeatParams fp;
  bool have_state = false, have_cov = false, have_batch = false;
  // feature batch
  float* uv = nullptr;
  int *clone_idx = nullptr, *n_meas = nullptr;
  double* p_FinG = nullptr;
  int n_feats = 0, max_meas = 0;
  // work buffers
  double *G = nullptr, *rec = nullptr, *chi2 = nullptr, *Bscr = nullptr;
  unsigned char* accept = nullptr;
  float* uvn = nullptr;            // normalised measurements for ovp_triangulate (allocated on first use)
  unsigned char* tri_ok = nullptr;
  int ldg = 0;
  double *gramS = nullptr, *gramR = nullptr, *part = nullptr, *Dinv = nullptr, *Ltp = nullptr;
  int n_chunks = 0, rows_per_chunk = 0, n_split = 0;
  double* Ab = nullptr;  // (n_max+1) x ld
  double *L = nullptr, *W1 = nullptr, *T = nullptr, *Lt = nullptr, *Y = nullptr;
  double* dx = nullptr;
  int* flags = nullptr;  // [0] not spd, [1] neg diag
  double *Hd = nullptr, *Acc = nullptr, *bcc = nullptr, *resd = nullptr;  // dense-H path
  // ovp_msckf_dense_blocks: the information pair of the accepted dense blocks over the union of their columns, waiting for the
  // point update of the same frame (added to Ab behind K2); empty = none
  std::vector<int> dense_cols;
  std::vector<double> dense_A, dense_b;
  void* slam_res = nullptr;        // ovp_slam_update: per-landmark [chi2 | status]
  double* slam_hscr = nullptr;     // ... blocks that do not fit LDS
  size_t slam_res_cap = 0, slam_hscr_cap = 0;
  double* dinit_buf = nullptr;     // ovp_slam_delayed_init: result blocks + shared scratch of the candidate loop
  size_t dinit_cap = 0;
  size_t Hd_cap = 0, res_cap = 0;
  int calib_id = -1, intr_id = -1;
  long long* dbg_cycles = nullptr;
  // plane path
  std::vector<int> h_n_meas, h_clone_idx;  // host copies of the uploaded batch (plane grouping is host logic)
  int *pl_featlist = nullptr, *pl_sid = nullptr;
  double *pl_cp = nullptr, *pl_cp_fej = nullptr, *pl_cst = nullptr, *pl_cstsum = nullptr, *pl_E = nullptr;
  double *pl_An = nullptr, *pl_bn = nullptr, *pl_Lr = nullptr, *pl_Dinv2 = nullptr, *pl_scal = nullptr;
  double *pl_res = nullptr, *pl_dx = nullptr;
  int pl_cap = 0;
  // SLAM landmarks on out-of-state planes (ovp_msckf_plane_update): [plane | id] ints and [p | p_fej] doubles
  int *pl_slam_i = nullptr;
  double *pl_slam_d = nullptr;
  int pl_slam_cap = 0, pl_n_slam = 0;
  // second-generation plane loop (k_plane2.hip / k_chol2.hip)
  double *pl_Tbuf = nullptr, *pl_crow = nullptr, *pl_dxlast = nullptr;
  int *pl_cur = nullptr, *pl_perm = nullptr;
  unsigned* pl_range_done = nullptr;
  unsigned pl_seq = 0;
  int range_lo = -1, range_hi = -1;   // ovp_batch_set_range (-1, -1 = whole batch)
  unsigned char* pl_used = nullptr;   // [f_max] features consumed by accepted planes (device)
  std::vector<unsigned char> h_pl_used;  // host copy of it behind the last plane loop (ovp_msckf_update_sharded splits the leftovers)
  int* h_slot = nullptr;              // [f_max] host-mapped: row block of a feature in the compacted rec / G of a point update (-1: none)
  int* d_slot = nullptr;              // its device address
  bool pl_used_valid = false;         // pl_used refers to the uploaded batch
  int pl2_cap = 0;
  // plane loop on a sub-state (n above the tile factorization's limit): accumulated pair, u rows, remapped id tables
  void *io_h = nullptr, *io_d = nullptr;         // ovp_io_arena: pinned host block + device block of the small entry points
  size_t io_cap = 0;
  double *pl_xbuf = nullptr, *pl_xy = nullptr;   // split plane solve: exported panels, [xzz(2) | y blocks]
  unsigned* pl_xflag = nullptr;                  // [32 step flags | 2 sync words]
  double *pl_Asum = nullptr, *pl_U = nullptr;
  int pl_U_cap = 0;
  void *pl_sub_tab = nullptr, *pl_sub_htab = nullptr;  // [ids | inverse | clone ids | column map] of the loop's column order
  bool pl_sub_active = false;   // ovp_msckf_plane_update runs inside plane_update_ordered (remapped tables, c->P = permuted copy)
  bool pl_sub_rest = false;     // ... on a marginal: the rest of the state follows by push-through (k_plane_sub_accum per plane)
  std::vector<int> pl_nl;       // [plane] leading columns involved up to and including that plane (loop order)
  double* pl_scatter_dst = nullptr;   // full order: where the covariance product of the loop is un-permuted to
  const int* pl_scatter_ids = nullptr;
  double pl_t_entry = 0.0;
  bool pl_psd = false;
  // A factor of the RESIDENT covariance left behind by the plane loop (P = V^T V, Lkeep = V^T in the state's column order): the point
  // update that follows needs some M with M M^T = P, not the Cholesky factor - chol(P) (the longer branch of the fused feature
  // launch at N = 240) is skipped.  Cleared by everything that writes P.
  double* Lkeep = nullptr;
  bool have_factor = false, use_kept_factor = false;
  double clone_jitter = 0.0;  // ovp_cov_clone_jitter: relative inflation of a cloned block's diagonal (0 = exact copy, the reference)
  double* boost_vec = nullptr;  // [n_max] k_gather_block_boost: the plane loop's diagonal boost by STATE column (zero where none)
  bool pl_boost_active = false, kept_boost = false;
  double* boost = nullptr;   // CholJob::boost: the amounts the reversed-order chol(P) added to the diagonal in front of the batch's columns
  int point_boost_n = 0;
  int point_nl = 0;  // > 0: chol(P) of the running point update was taken in reversed index order (CholJob::flip) and the update's
                     // T = I + L^T A L is the identity outside its leading point_nl columns          // second attempt of a plane loop whose chol(P) failed: pivot-dropping factor of the PSD prior
  hipEvent_t ev_subtab = nullptr;     // behind the upload of pl_sub_htab (the pinned block is rewritten by the next call)
  void *pl_hstage = nullptr, *pl_dstage = nullptr;  // pinned host / device staging of the per-call tables
  size_t pl_stage_cap = 0;
  void* pl_hres = nullptr;            // pinned host copy of the plane results
  void* pl_hres_dev = nullptr;        // ... its device address (mapped: the plane loop publishes its results into it from a kernel)
  unsigned pl_pub_seq = 0;            // sequence number of the plane loop's last publication
  size_t pl_hres_cap = 0;
  // one device block + one pinned staging block each for the pose tables and for the feature batch (a single copy per upload)
  void *state_block = nullptr, *h_state_stage = nullptr, *batch_block = nullptr, *h_batch_stage = nullptr;
  size_t state_bytes = 0, batch_cap = 0;
  size_t so_R = 0, so_Rf = 0, so_p = 0, so_pf = 0, so_cal = 0, so_id = 0, so_cm = 0;
  hipEvent_t ev_state = nullptr, ev_batch = nullptr;
  int pl_ktimer = 0;  // 1 = events around every k_chol2 launch and around the loop, 2 = around the loop only
  std::vector<hipEvent_t> pl_ev, pl_ev_loop;
  double pl_ktime_ms = 0.0;
  int pl_klaunches = 0;
  int* idbuf = nullptr;      // scratch ints (ids)
  double* smallbuf = nullptr;  // scratch doubles (Phi, Q, CPT, PCP, marginal)
  size_t small_cap = 0;
  // sub-state update (n above the tile factorization's limit): involved state columns and six ns x ns scratch matrices
  int* sub_ids = nullptr;
  int sub_ns = 0;
  std::vector<int> h_clone_id;  // host copy of the clone columns (ovp_state_upload)
  double* sub_buf = nullptr;
  // pinned host staging
  double *h_dx = nullptr, *h_chi2 = nullptr;
  unsigned char* h_accept = nullptr;
  int* h_flags = nullptr;
  void *res_block = nullptr, *h_res_block = nullptr;  // [flags | dx | chi2 | accept], device and pinned host
  void* h_res_block_dev = nullptr;                    // device address of the pinned block
  volatile unsigned* h_seq = nullptr;                 // sequence word behind it (written last by k_publish_results)
  unsigned prop_seq = 0;                              // ovp_cov_propagate's own sequence (words [4], [5] behind h_seq: seq, verdict)
  unsigned seq = 0, pub_seq = 0;
  bool pub_pending = false;      // the running update publishes its results itself (k_dx_rows)
  bool need_join = false;     // chol(P) / K2 of the current update finish on stream2 (ev_join) rather than on the main stream
  unsigned* ticket = nullptr;    // block counter of the publishing kernel
  std::vector<int> h_nmeas;                           // host copy of n_meas of the current batch (row count of `info`)
  bool h_nmeas_valid = false;
  size_t res_bytes = 0;
  float last_ms[4] = {0, 0, 0, 0};
  bool timed = false;
  // dominant-kernel timer
  bool ktimer = false;
  hipEvent_t ev_k0 = nullptr, ev_k1 = nullptr;
  double ktime_ms = 0.0;
  int klaunches = 0;
  bool kpending = false;
  // host-side clock of the two update entry points, accumulated (ovp_host_timing): plane loop [entry -> first launch | entry -> last
  // launch enqueued | wait for the device | calls], point update [enqueue | wait | calls]
  double host_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};

// Everything that writes the covariance calls this: the factor the plane loop left (Lkeep) and the bookkeeping of a staged point
// update that was built but never applied (use_kept_factor / point_nl, set by ovp_msckf_build_gate_gram_async) no longer belong to P.
static inline void drop_kept_factor(ovp_ctx* c) {
  if (!c) return;
  c->have_factor = false;
  c->use_kept_factor = false;
  c->point_nl = 0;
  c->point_boost_n = 0;
  c->kept_boost = false;
  c->dense_cols.clear();  // (a pending dense pair was gated against the covariance that is being replaced)
}

static inline double host_now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}


template <class T>
static hipError_t dalloc(T** p, size_t count) {
  return hipMalloc((void**)p, count * sizeof(T));
}

// ---- shared between the entry-point files ------------------------------------------------------------------------------------
extern "C" int ovp_io_arena(ovp_ctx* c, size_t bytes, void** host, void** dev);  // pinned staging arena (ovp_api_ctx.hip)
// ovp_api_point.hip
int fill_feat_params(ovp_ctx* c, const ovp_update_opts* o);
int ovp_fetch_to_hres(ovp_ctx* c, const void* dsrc, size_t bytes, hipStream_t s);  // device block -> c->pl_hres, waited for (ovp_api_plane.hip)
int chol_of_P(ovp_ctx* c, hipStream_t s);
hipError_t chol_of_T(ovp_ctx* c, const double* T, int n, int ld, int add_identity, const int* cond, hipStream_t s);
int set_substate(ovp_ctx* c, const std::vector<int>& ids);
int ekf_from_gram(ovp_ctx* c, bool chol_p_done_on_stream2, bool publish = false);
int ekf_sform(ovp_ctx* c);
// ovp_api_plane.hip
int plane2_buffers(ovp_ctx* c, int NP, size_t stage_bytes, size_t res_bytes);
