// C-ABI shim of libovplane_hip.so (see include/ovplane_hip.h). Host-side orchestration only: every arithmetic
// step of the update path runs in the gfx950 kernels of k_feat.hip / k_gram.hip / k_ekf.hip.
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
// chi-square 0.95 quantile (replaces boost::math::quantile(chi_squared(k), 0.95),
// update/UpdaterMSCKF.cpp:59-62,749-750): regularised incomplete gamma + safeguarded Newton.
// ------------------------------------------------------------------------------------------------
static double gammap_reg(double a, double x) {
  if (x <= 0) return 0.0;
  const double gln = lgamma(a);
  if (x < a + 1.0) {
    double ap = a, sum = 1.0 / a, del = sum;
    for (int n = 0; n < 100000; ++n) {
      ap += 1.0;
      del *= x / ap;
      sum += del;
      if (fabs(del) < fabs(sum) * 1e-17) break;
    }
    return sum * exp(-x + a * log(x) - gln);
  }
  double b = x + 1.0 - a, c = 1.0 / 1e-300, d = 1.0 / b, h = d;
  for (int i = 1; i < 100000; ++i) {
    const double an = -i * (i - a);
    b += 2.0;
    d = an * d + b;
    if (fabs(d) < 1e-300) d = 1e-300;
    c = b + an / c;
    if (fabs(c) < 1e-300) c = 1e-300;
    d = 1.0 / d;
    const double del = d * c;
    h *= del;
    if (fabs(del - 1.0) < 1e-17) break;
  }
  return 1.0 - exp(-x + a * log(x) - gln) * h;
}

extern "C" double ovp_chi2_quantile_095(int dof) {
  if (dof < 1) return 0.0;
  const double p = 0.95, a = 0.5 * (double)dof;
  const double z = 1.6448536269514722;
  const double t = 1.0 - 2.0 / (9.0 * dof) + z * sqrt(2.0 / (9.0 * dof));
  double x = 0.5 * dof * t * t * t;
  if (x <= 0) x = 0.5;
  double lo = 0.0, hi = 1e300;
  for (int it = 0; it < 200; ++it) {
    const double f = gammap_reg(a, x) - p;
    if (f > 0) hi = x; else lo = x;
    const double dens = exp((a - 1.0) * log(x) - x - lgamma(a));
    double xn = x - f / dens;
    if (!(xn > lo && xn < hi)) xn = (hi < 1e299) ? 0.5 * (lo + hi) : 2.0 * x;
    if (fabs(xn - x) <= 1e-15 * fabs(xn)) {
      x = xn;
      break;
    }
    x = xn;
  }
  return 2.0 * x;
}

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
}

static inline double host_now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

extern "C" const char* ovp_version(void) { return "ovplane_hip 0.5 (gfx950)"; }

extern "C" const char* ovp_error_string(int code) {
  switch (code) {
    case 0: return "ok";
    case OVP_E_ARG: return "bad argument";
    case OVP_E_CAPACITY: return "capacity exceeded";
    case OVP_E_NOTSPD: return "matrix not positive definite";
    case OVP_E_NEGDIAG: return "negative covariance diagonal";
    case OVP_E_NODEVICE: return "no usable HIP device";
    case OVP_E_STATE: return "call order violated";
    case OVP_E_TIMEOUT: return "device-side hand-over timed out (workgroups of the plane solve not co-resident)";
    case OVP_E_RCCL: return "RCCL not loadable or a collective call failed";
    case OVP_E_PEER: return "sharded update: the build of another rank failed (errors are collective)";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown";
  }
}

template <class T>
static hipError_t dalloc(T** p, size_t count) {
  return hipMalloc((void**)p, count * sizeof(T));
}

extern "C" int ovp_ctx_create(int device, int n_state_max, int n_clones_max, int n_feats_max, void* stream,
                              ovp_ctx** out) {
  if (!out || n_state_max < 1 || n_clones_max < 1 || n_clones_max > OVP_MAX_CLONES || n_feats_max < 1) return OVP_E_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= device) return OVP_E_NODEVICE;
  HIPCHK(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return OVP_E_NODEVICE;  // gfx950-only build, no fallback
  ovp_ctx* c = new ovp_ctx();
  c->device = device;
  // Two streams: the main one carries K1/K2/K3, the side stream the measurement-independent chol(P).  (Pinning the side
  // stream to one CU with hipExtStreamCreateWithCUMask was tried: the driver keeps CU masks symmetric across shader
  // engines, so removing one CU from the main stream removes 32 and K1 drops below one block per feature.)
  if (stream) {
    c->stream = (hipStream_t)stream;
  } else {
    HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->own_stream = true;
  }
  HIPCHK(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
  HIPCHK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
  for (int i = 0; i < 6; ++i) HIPCHK(hipEventCreate(&c->ev_t[i]));
  HIPCHK(hipEventCreate(&c->ev_k0));
  HIPCHK(hipEventCreate(&c->ev_k1));
  c->n_max = n_state_max;
  c->c_max = n_clones_max;
  c->f_max = n_feats_max;
  c->ld = round_up(n_state_max, 16);
  c->ldg = round_up(n_state_max + 4, 16);  // state columns | residual | 3 out-of-state plane columns
  if (c->ldg > OVP_LDG_CAP) return OVP_E_CAPACITY;  // the feature kernels stage 3 projector rows in LDS  // K1 stages the projector rows in its 64x65/2 LDS triangle
  const size_t nn = (size_t)(c->n_max + 1) * c->ld;
  HIPCHK(dalloc(&c->P, nn));
  HIPCHK(dalloc(&c->P_tmp, nn));
  HIPCHK(dalloc(&c->Ab, nn + 8));  // (+ the peer-error word the sharded update sums along with the pair)
  HIPCHK(dalloc(&c->L, nn));
  HIPCHK(dalloc(&c->W1, nn));
  HIPCHK(dalloc(&c->T, nn));
  HIPCHK(dalloc(&c->Lt, nn));
  HIPCHK(dalloc(&c->Y, nn));
  // results of an update live in ONE block [flags 4 x i32 | dx n_max | chi2 f_max | accept f_max] so that
  // ovp_msckf_fetch_results is a single device-to-host copy (four small copies cost ~5 us each)
  c->res_bytes = 16 + sizeof(double) * ((size_t)c->n_max + n_feats_max) + (size_t)n_feats_max;
  HIPCHK(hipMalloc((void**)&c->res_block, c->res_bytes));
  HIPCHK(hipMemset(c->res_block, 0, c->res_bytes));
  c->flags = (int*)c->res_block;
  c->dx = (double*)((char*)c->res_block + 16);
  c->chi2 = c->dx + c->n_max;
  c->accept = (unsigned char*)(c->chi2 + n_feats_max);
  {
    // pose tables: [R | R_fej | p | p_fej | cal(32) | clone_id | colmap], fixed offsets (capacities), uploaded as one block
    auto al = [](size_t v) { return (v + 63) & ~(size_t)63; };
    size_t o = 0;
    c->so_R = o;
    o = al(o + sizeof(double) * 9 * n_clones_max);
    c->so_Rf = o;
    o = al(o + sizeof(double) * 9 * n_clones_max);
    c->so_p = o;
    o = al(o + sizeof(double) * 3 * n_clones_max);
    c->so_pf = o;
    o = al(o + sizeof(double) * 3 * n_clones_max);
    c->so_cal = o;
    o = al(o + sizeof(double) * 32);
    c->so_id = o;
    o = al(o + sizeof(int) * n_clones_max);
    c->so_cm = o;
    o = al(o + sizeof(ovp::ColMap) * c->n_max);
    c->state_bytes = o;
    HIPCHK(hipMalloc(&c->state_block, o));
    HIPCHK(hipMemset(c->state_block, 0, o));
    HIPCHK(hipHostMalloc(&c->h_state_stage, o, hipHostMallocDefault));
    char* b = (char*)c->state_block;
    c->clone_R = (double*)(b + c->so_R);
    c->clone_R_fej = (double*)(b + c->so_Rf);
    c->clone_p = (double*)(b + c->so_p);
    c->clone_p_fej = (double*)(b + c->so_pf);
    c->cal = (double*)(b + c->so_cal);
    c->clone_id = (int*)(b + c->so_id);
    c->colmap = (ovp::ColMap*)(b + c->so_cm);
    HIPCHK(hipEventCreateWithFlags(&c->ev_state, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->ev_batch, hipEventDisableTiming));
  }
  HIPCHK(dalloc(&c->chi2_table, (size_t)OVP_CHI2_TABLE + 1));
  {
    // feature batch: [p_FinG | n_meas | clone_idx | uv], compact per upload (p_FinG always first: ovp_triangulate writes it)
    const size_t F = (size_t)n_feats_max, M = OVP_MAX_MEAS;
    c->batch_cap = sizeof(double) * 3 * F + sizeof(int) * F + sizeof(int) * F * M + sizeof(float) * 2 * F * M + 256;
    HIPCHK(hipMalloc(&c->batch_block, c->batch_cap));
    HIPCHK(hipHostMalloc(&c->h_batch_stage, c->batch_cap, hipHostMallocDefault));
    c->p_FinG = (double*)c->batch_block;
  }
  HIPCHK(dalloc(&c->G, (size_t)3 * n_feats_max * c->ldg));
  HIPCHK(dalloc(&c->Bscr, (size_t)n_feats_max * OVP_BSCR));
  HIPCHK(dalloc(&c->rec, (size_t)n_clones_max * n_feats_max * 2 * 21));
  // reduction geometry: fixed per context so the summation order (hence the result bits) is reproducible
  c->rows_per_chunk = 128;  // 32 rows = 8 MFMA steps per wave of k_gram_pair
  c->n_chunks = (2 * n_feats_max + c->rows_per_chunk - 1) / c->rows_per_chunk;
  HIPCHK(dalloc(&c->gramS, (size_t)n_clones_max * c->n_chunks * OVP_GRAM_ELEMS));
  HIPCHK(dalloc(&c->gramR, (size_t)n_clones_max * OVP_GRAM_ELEMS));
  HIPCHK(dalloc(&c->Dinv, (size_t)(c->ld / 16 + 1) * 256));
  {
    const size_t ntm = (size_t)c->ld / 16 + 1;
    HIPCHK(dalloc(&c->Ltp, ntm * (ntm + 1) / 2 * 256));  // tile-packed factor for k_fwdsub
  }
  c->n_split = (3 * n_feats_max + 63) / 64;  // split-K partials of the dense Gram product (k_gram_pair: ~256 blocks)
  if (c->n_split < 1) c->n_split = 1;
  if (c->n_split > 64) c->n_split = 64;
  {
    const int nt = c->ldg / 16;
    HIPCHK(dalloc(&c->part, (size_t)c->n_split * (nt * (nt + 1) / 2) * 256));
  }
  HIPCHK(dalloc(&c->idbuf, (size_t)4 * c->n_max + 64));
  c->small_cap = (size_t)4 * c->n_max * 64 + (size_t)c->n_max * c->n_max;
  HIPCHK(dalloc(&c->smallbuf, c->small_cap));
  HIPCHK(hipMalloc((void**)&c->ticket, 16));
  HIPCHK(hipMemset(c->ticket, 0, 16));
  HIPCHK(hipHostMalloc((void**)&c->h_res_block, c->res_bytes + 64, hipHostMallocMapped));  // pinned mirror of res_block
  memset(c->h_res_block, 0, c->res_bytes + 64);
  HIPCHK(hipHostGetDevicePointer(&c->h_res_block_dev, c->h_res_block, 0));
  HIPCHK(hipHostMalloc((void**)&c->h_slot, sizeof(int) * (size_t)(n_feats_max + 16), hipHostMallocMapped));
  HIPCHK(hipHostGetDevicePointer((void**)&c->d_slot, c->h_slot, 0));
  c->h_seq = (volatile unsigned*)((char*)c->h_res_block + ((c->res_bytes + 15) & ~(size_t)15));
  c->h_flags = (int*)c->h_res_block;
  c->h_dx = (double*)((char*)c->h_res_block + 16);
  c->h_chi2 = c->h_dx + c->n_max;
  c->h_accept = (unsigned char*)(c->h_chi2 + n_feats_max);
  // chi2 table
  {
    std::vector<double> tab(OVP_CHI2_TABLE + 1, 0.0);
    for (int k = 1; k <= OVP_CHI2_TABLE; ++k) tab[k] = ovp_chi2_quantile_095(k);
    HIPCHK(hipMemcpy(c->chi2_table, tab.data(), sizeof(double) * tab.size(), hipMemcpyHostToDevice));
  }
  memset(&c->fp, 0, sizeof(c->fp));
  *out = c;
  return 0;
}

extern "C" int ovp_ctx_destroy(ovp_ctx* c) {
  if (!c) return OVP_E_ARG;
  hipSetDevice(c->device);
  hipStreamSynchronize(c->stream);
  hipStreamSynchronize(c->stream2);
  void* dev[] = {c->P, c->P_tmp, c->Ab, c->L, c->W1, c->T, c->Lt, c->Y, c->res_block, c->ticket, c->uvn, c->tri_ok, c->state_block, c->batch_block,
                 c->chi2_table, c->G, c->Bscr, c->rec, c->gramS, c->gramR, c->Dinv, c->Ltp, c->part, c->idbuf, c->smallbuf, c->Hd, c->Acc,
                 c->bcc, c->resd, c->pl_slam_i, c->pl_slam_d, c->sub_ids, c->sub_buf, c->pl_Tbuf, c->pl_crow, c->pl_dxlast,
                 c->pl_cur, c->pl_perm, c->pl_range_done, c->pl_used, c->pl_dstage, c->pl_xbuf, c->pl_xy, c->pl_xflag, c->pl_Asum,
                 c->pl_U, c->pl_sub_tab, c->Lkeep, c->slam_res, c->slam_hscr, c->dinit_buf, c->boost, c->boost_vec};
  for (void* p : dev)
    if (p) hipFree(p);
  if (c->h_res_block) hipHostFree(c->h_res_block);
  if (c->h_slot) hipHostFree(c->h_slot);
  if (c->h_state_stage) hipHostFree(c->h_state_stage);
  if (c->h_batch_stage) hipHostFree(c->h_batch_stage);
  if (c->ev_state) hipEventDestroy(c->ev_state);
  if (c->ev_batch) hipEventDestroy(c->ev_batch);
  if (c->pl_hstage) hipHostFree(c->pl_hstage);
  if (c->pl_hres) hipHostFree(c->pl_hres);
  if (c->pl_sub_htab) hipHostFree(c->pl_sub_htab);
  if (c->ev_subtab) hipEventDestroy(c->ev_subtab);
  for (hipEvent_t e : c->pl_ev) hipEventDestroy(e);
  for (hipEvent_t e : c->pl_ev_loop) hipEventDestroy(e);
  if (c->io_h) hipHostFree(c->io_h);
  if (c->io_d) hipFree(c->io_d);
  hipEventDestroy(c->ev_fork);
  hipEventDestroy(c->ev_join);
  for (int i = 0; i < 6; ++i) hipEventDestroy(c->ev_t[i]);
  hipEventDestroy(c->ev_k0);
  hipEventDestroy(c->ev_k1);
  hipStreamDestroy(c->stream2);
  if (c->own_stream) hipStreamDestroy(c->stream);
  delete c;
  return 0;
}

extern "C" int ovp_sync(ovp_ctx* c) {
  if (!c) return OVP_E_ARG;
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

// Results go to the host without a copy command: the last kernel of an update writes the result block into mapped pinned
// memory and then a sequence number; ovp_msckf_fetch_results spins on that word (a hipMemcpyAsync + hipStreamSynchronize
// pair costs ~25 us of launch, blit and wake-up latency per update, this ~5).
__global__ __launch_bounds__(1024) void k_publish_results(unsigned long long* __restrict__ src,
                                                         unsigned long long* __restrict__ dst, int words,
                                                         volatile unsigned* seq_host, unsigned seq) {
  for (int i = threadIdx.x; i < words; i += 1024) dst[i] = src[i];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) *seq_host = seq;
  if (threadIdx.x < 2) src[threadIdx.x] = 0ull;  // the four flag words, cleared for the next update
}

extern "C" int ovp_ctx_stream(ovp_ctx* c, void** stream) {
  if (!c || !stream) return OVP_E_ARG;
  *stream = (void*)c->stream;
  return 0;
}

extern "C" int ovp_cov_size(ovp_ctx* c) { return c ? c->n : OVP_E_ARG; }

// ---- covariance residency ----------------------------------------------------------------------
extern "C" int ovp_io_arena(ovp_ctx* c, size_t bytes, void** host, void** dev);

// (upload / download / marginal / propagate go through the pinned arena: one contiguous copy each way.  A 2-D copy from
//  pageable memory cost 90 us of host time at N = 130, a pageable copy per small array 8 us each.)
extern "C" int ovp_cov_upload(ovp_ctx* c, const double* P_host, int n, int ld) {
  drop_kept_factor(c);  // (writes the covariance: a kept factor no longer belongs to it)
  if (!c || !P_host || n < 1 || ld < n) return OVP_E_ARG;
  if (n > c->n_max) return OVP_E_CAPACITY;
  void *ah = nullptr, *ad = nullptr;
  const size_t bytes = sizeof(double) * (size_t)n * c->ld;
  {
    const int rca = ovp_io_arena(c, bytes, &ah, &ad);
    if (rca) return rca;
  }
  for (int i = 0; i < n; ++i) {
    memcpy((double*)ah + (size_t)i * c->ld, P_host + (size_t)i * ld, sizeof(double) * n);
    if (c->ld > n) memset((double*)ah + (size_t)i * c->ld + n, 0, sizeof(double) * (c->ld - n));
  }
  HIPCHK(hipMemcpyAsync(c->P, ah, bytes, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  c->n = n;
  c->have_cov = true;
  return 0;
}
extern "C" int ovp_cov_set_device(ovp_ctx* c, const double* P_dev, int n, int ld) {
  drop_kept_factor(c);  // (writes the covariance: a kept factor no longer belongs to it)
  if (!c || !P_dev || n < 1 || ld < n) return OVP_E_ARG;
  if (n > c->n_max) return OVP_E_CAPACITY;
  HIPCHK(hipMemcpy2DAsync(c->P, sizeof(double) * c->ld, P_dev, sizeof(double) * ld, sizeof(double) * n, n,
                          hipMemcpyDeviceToDevice, c->stream));
  c->n = n;
  c->have_cov = true;
  return 0;
}
extern "C" int ovp_cov_download(ovp_ctx* c, double* P_host, int n, int ld) {
  if (!c || !P_host || n != c->n || ld < n) return OVP_E_ARG;
  void *ah = nullptr, *ad = nullptr;
  const size_t bytes = sizeof(double) * (size_t)n * c->ld;
  {
    const int rca = ovp_io_arena(c, bytes, &ah, &ad);
    if (rca) return rca;
  }
  HIPCHK(hipMemcpyAsync(ah, c->P, bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  for (int i = 0; i < n; ++i) memcpy(P_host + (size_t)i * ld, (const double*)ah + (size_t)i * c->ld, sizeof(double) * n);
  return 0;
}
extern "C" int ovp_cov_marginal(ovp_ctx* c, const int* ids, const int* sizes, int n_vars, double* out_host) {
  if (!c || !ids || !sizes || !out_host || n_vars < 1) return OVP_E_ARG;
  std::vector<int> cols;
  for (int i = 0; i < n_vars; ++i)
    for (int k = 0; k < sizes[i]; ++k) {
      if (ids[i] + k >= c->n || ids[i] < 0) return OVP_E_ARG;
      cols.push_back(ids[i] + k);
    }
  const int m = (int)cols.size();
  if (m > c->n_max || (size_t)m * m > c->small_cap) return OVP_E_CAPACITY;
  void *ah = nullptr, *ad = nullptr;
  const size_t o_out = ((sizeof(int) * (size_t)m + 63) / 64) * 64, bytes = o_out + sizeof(double) * (size_t)m * m;
  {
    const int rca = ovp_io_arena(c, bytes, &ah, &ad);
    if (rca) return rca;
  }
  memcpy(ah, cols.data(), sizeof(int) * m);
  HIPCHK(hipMemcpyAsync(ad, ah, sizeof(int) * m, hipMemcpyHostToDevice, c->stream));
  HIPCHK(ovp_launch_gather_marginal(c->P, c->ld, (const int*)ad, m, (double*)((char*)ad + o_out), c->stream));
  HIPCHK(hipMemcpyAsync((char*)ah + o_out, (char*)ad + o_out, sizeof(double) * (size_t)m * m, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  memcpy(out_host, (char*)ah + o_out, sizeof(double) * (size_t)m * m);
  return 0;
}

// ---- state tables ------------------------------------------------------------------------------
static void quat_2_rot(const double q[4], double R[9]) {
  // JPL: R = (2 q4^2 - 1) I - 2 q4 [qv]x + 2 qv qv^T  (ext quat_ops.h)
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double a = 2.0 * w * w - 1.0;
  R[0] = a + 2.0 * x * x;
  R[1] = 2.0 * w * z + 2.0 * x * y;
  R[2] = -2.0 * w * y + 2.0 * x * z;
  R[3] = -2.0 * w * z + 2.0 * y * x;
  R[4] = a + 2.0 * y * y;
  R[5] = 2.0 * w * x + 2.0 * y * z;
  R[6] = 2.0 * w * y + 2.0 * z * x;
  R[7] = -2.0 * w * x + 2.0 * z * y;
  R[8] = a + 2.0 * z * z;
}

extern "C" int ovp_state_upload(ovp_ctx* c, const ovp_state_tables* st) {
  if (!c || !st || !st->clone_q || !st->clone_p || !st->clone_q_fej || !st->clone_p_fej || !st->clone_id) return OVP_E_ARG;
  if (st->n_clones < 1 || st->n_clones > c->c_max || st->n_state > c->n_max) return OVP_E_CAPACITY;
  const int C = st->n_clones;
  for (int i = 0; i < C; ++i)
    if (st->clone_id[i] < 0 || st->clone_id[i] + 6 > st->n_state) return OVP_E_ARG;
  // everything goes through ONE pinned block and ONE copy, no synchronisation (seven pageable copies + a sync cost ~50 us,
  // more than the GPU time of a small update)
  HIPCHK(hipEventSynchronize(c->ev_state));  // the previous upload has left the staging block (normally long ago)
  char* h = (char*)c->h_state_stage;
  double* R = (double*)(h + c->so_R);
  double* Rf = (double*)(h + c->so_Rf);
  for (int i = 0; i < C; ++i) {
    quat_2_rot(st->clone_q + 4 * i, R + 9 * i);
    quat_2_rot(st->clone_q_fej + 4 * i, Rf + 9 * i);
  }
  memcpy(h + c->so_p, st->clone_p, sizeof(double) * 3 * C);
  memcpy(h + c->so_pf, st->clone_p_fej, sizeof(double) * 3 * C);
  memcpy(h + c->so_id, st->clone_id, sizeof(int) * C);
  {
    double* cal = (double*)(h + c->so_cal);
    quat_2_rot(st->calib_q, cal);
    memcpy(cal + 9, st->calib_p, sizeof(double) * 3);
    memcpy(cal + 12, st->intrinsics, sizeof(double) * 8);
  }
  // column map for the assembly kernel (calibration columns are enabled per update via the opts)
  ovp::ColMap* cm = (ovp::ColMap*)(h + c->so_cm);
  memset(cm, 0, sizeof(ovp::ColMap) * c->n_max);
  for (int i = 0; i < C; ++i)
    for (int k = 0; k < 6; ++k) {
      ovp::ColMap& m = cm[st->clone_id[i] + k];
      m.kind = 1;
      m.idx = i;
      m.off = k;
    }
  if (st->calib_id >= 0 && st->calib_id + 6 <= st->n_state)
    for (int k = 0; k < 6; ++k) {
      ovp::ColMap& m = cm[st->calib_id + k];
      m.kind = 2;
      m.idx = k;
    }
  if (st->intr_id >= 0 && st->intr_id + 8 <= st->n_state)
    for (int k = 0; k < 8; ++k) {
      ovp::ColMap& m = cm[st->intr_id + k];
      m.kind = 2;
      m.idx = 6 + k;
    }
  HIPCHK(hipMemcpyAsync(c->state_block, c->h_state_stage, c->state_bytes, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipEventRecord(c->ev_state, c->stream));
  ovp::FeatParams& fp = c->fp;
  fp.clone_R = c->clone_R;
  fp.clone_p = c->clone_p;
  fp.clone_R_fej = c->clone_R_fej;
  fp.clone_p_fej = c->clone_p_fej;
  fp.clone_id = c->clone_id;
  fp.n_clones = C;
  fp.cal = c->cal;
  c->calib_id = st->calib_id;
  c->intr_id = st->intr_id;
  c->h_clone_id.assign(st->clone_id, st->clone_id + C);
  c->fp.fisheye = st->cam_fisheye ? 1 : 0;
  c->have_state = true;
  return 0;
}

// One pinned host block + one device block per context for the entry points whose arguments are a handful of small host arrays
// (triangulation, plane fitting, plane refinement): the inputs are packed into the host block and cross the bus in ONE copy, the
// outputs come back in one.  The first versions issued a pageable copy per array (137 copy kernels per closed-loop frame with
// planes, a quarter of its GPU time) and, in the plane-fitting entries, a hipMalloc / hipFree pair per call.
extern "C" int ovp_io_arena(ovp_ctx* c, size_t bytes, void** host, void** dev) {
  if (!c || !host || !dev) return OVP_E_ARG;
  if (bytes > c->io_cap) {
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->io_h) hipHostFree(c->io_h);
    if (c->io_d) hipFree(c->io_d);
    c->io_h = c->io_d = nullptr;
    c->io_cap = 0;
    const size_t cap = bytes + bytes / 2 + 4096;
    HIPCHK(hipHostMalloc(&c->io_h, cap, hipHostMallocDefault));
    HIPCHK(hipMalloc(&c->io_d, cap));
    c->io_cap = cap;
  }
  *host = c->io_h;
  *dev = c->io_d;
  return 0;
}

// ---- triangulation ----------------------------------------------------------------
extern "C" void ovp_triang_defaults(ovp_triang_opts* o) {
  if (!o) return;
  o->refine_features = 1;
  o->max_runs = 5;
  o->init_lamda = 1e-3;
  o->max_lamda = 1e10;
  o->min_dx = 1e-6;
  o->min_dcost = 1e-6;
  o->lam_mult = 10.0;
  o->min_dist = 0.10;
  o->max_dist = 60.0;
  o->max_baseline = 40.0;
  o->max_cond_number = 10000.0;
  o->triangulate_1d = 0;
  o->reserved = 0;
}

extern "C" int ovp_triangulate(ovp_ctx* c, const ovp_triang_opts* o, const float* uv_norm, double* p_FinG_out, uint8_t* ok) {
  if (!c || !o || !uv_norm || !ok) return OVP_E_ARG;
  if (!c->have_state || !c->have_batch) return OVP_E_STATE;
  const size_t F = (size_t)c->n_feats, M = (size_t)c->max_meas;
  if (F == 0) return 0;
  // arena: [uv_norm | -> p_FinG | ok]
  auto al = [](size_t v) { return (v + 63) & ~(size_t)63; };
  const size_t b_uv = sizeof(float) * F * M * 2, o_p = al(b_uv), o_ok = al(o_p + sizeof(double) * 3 * F), total = al(o_ok + F);
  void *ah = nullptr, *ad = nullptr;
  {
    const int rca = ovp_io_arena(c, total, &ah, &ad);
    if (rca) return rca;
  }
  memcpy(ah, uv_norm, b_uv);
  HIPCHK(hipMemcpyAsync(ad, ah, b_uv, hipMemcpyHostToDevice, c->stream));
  ovp::TriParams tp;
  tp.uvn = (const float*)ad;
  tp.clone_idx = c->fp.clone_idx;
  tp.n_meas = c->fp.n_meas;
  tp.n_feats = (int)F;
  tp.max_meas = (int)M;
  tp.clone_R = c->clone_R;
  tp.clone_p = c->clone_p;
  tp.cal = c->cal;
  tp.refine_features = o->refine_features;
  tp.triangulate_1d = o->triangulate_1d;
  tp.max_runs = o->max_runs;
  tp.init_lamda = o->init_lamda;
  tp.max_lamda = o->max_lamda;
  tp.min_dx = o->min_dx;
  tp.min_dcost = o->min_dcost;
  tp.lam_mult = o->lam_mult;
  tp.min_dist = o->min_dist;
  tp.max_dist = o->max_dist;
  tp.max_baseline = o->max_baseline;
  tp.max_cond_number = o->max_cond_number;
  tp.p_FinG = c->p_FinG;  // the library's own buffer even when the batch was bound to caller memory
  tp.ok = (unsigned char*)ad + o_ok;
  HIPCHK(ovp_launch_triangulate(&tp, c->stream));
  c->fp.p_FinG = c->p_FinG;
  // results into the pinned block (the positions stay in the batch's buffer on the device as linearisation points)
  HIPCHK(hipMemcpyAsync((char*)ah + o_p, c->p_FinG, sizeof(double) * 3 * F, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync((char*)ah + o_ok, (char*)ad + o_ok, F, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (p_FinG_out) memcpy(p_FinG_out, (char*)ah + o_p, sizeof(double) * 3 * F);
  memcpy(ok, (char*)ah + o_ok, F);
  return 0;
}

// ---- feature batch -----------------------------------------------------------------------------
extern "C" int ovp_batch_upload(ovp_ctx* c, const ovp_feature_batch* b) {
  if (!c || !b || b->n_feats < 0 || b->max_meas < 1 || b->max_meas > OVP_MAX_MEAS) return OVP_E_ARG;
  if (b->n_feats > c->f_max) return OVP_E_CAPACITY;
  const size_t F = (size_t)b->n_feats, M = (size_t)b->max_meas;
  // compact layout [p_FinG | n_meas | clone_idx | uv] in one pinned block, one copy, no synchronisation: the caller's arrays
  // are free again on return because they were copied into the staging block
  auto al = [](size_t v) { return (v + 63) & ~(size_t)63; };
  const size_t o_p = 0, o_nm = al(o_p + sizeof(double) * 3 * F), o_ci = al(o_nm + sizeof(int) * F),
               o_uv = al(o_ci + sizeof(int) * F * M), total = al(o_uv + sizeof(float) * 2 * F * M);
  if (total > c->batch_cap) return OVP_E_CAPACITY;
  char* d = (char*)c->batch_block;
  c->p_FinG = (double*)(d + o_p);
  c->n_meas = (int*)(d + o_nm);
  c->clone_idx = (int*)(d + o_ci);
  c->uv = (float*)(d + o_uv);
  if (F) {
    HIPCHK(hipEventSynchronize(c->ev_batch));
    char* h = (char*)c->h_batch_stage;
    memcpy(h + o_p, b->p_FinG, sizeof(double) * 3 * F);
    memcpy(h + o_nm, b->n_meas, sizeof(int) * F);
    memcpy(h + o_ci, b->clone_idx, sizeof(int) * F * M);
    memcpy(h + o_uv, b->uv, sizeof(float) * 2 * F * M);
    HIPCHK(hipMemcpyAsync(c->batch_block, c->h_batch_stage, total, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipEventRecord(c->ev_batch, c->stream));
  }
  c->h_n_meas.assign(b->n_meas, b->n_meas + F);
  c->h_nmeas.assign(b->n_meas, b->n_meas + F);
  c->h_nmeas_valid = true;
  c->h_clone_idx.assign(b->clone_idx, b->clone_idx + F * M);
  c->fp.uv = c->uv;
  c->fp.clone_idx = c->clone_idx;
  c->fp.n_meas = c->n_meas;
  c->fp.p_FinG = c->p_FinG;
  c->n_feats = b->n_feats;
  c->max_meas = b->max_meas;
  c->have_batch = true;
  c->pl_used_valid = false;
  c->range_lo = c->range_hi = -1;
  return 0;
}
extern "C" int ovp_batch_set_range(ovp_ctx* c, int lo, int hi) {
  if (!c) return OVP_E_ARG;
  if (!c->have_batch) return OVP_E_STATE;
  if (lo == -1 && hi == -1) {
    c->range_lo = c->range_hi = -1;
    return 0;
  }
  if (lo < 0 || hi < 0 || hi > c->n_feats) return OVP_E_ARG;
  c->range_lo = lo;
  c->range_hi = hi < lo ? lo : hi;
  return 0;
}
extern "C" int ovp_batch_bind_device(ovp_ctx* c, const ovp_feature_batch* b) {
  if (!c || !b || b->n_feats < 0 || b->max_meas < 1 || b->max_meas > OVP_MAX_MEAS) return OVP_E_ARG;
  if (b->n_feats > c->f_max) return OVP_E_CAPACITY;
  c->h_n_meas.clear();
  c->h_nmeas_valid = false;  // read back lazily (once) if a caller asks for ovp_update_info
  c->h_clone_idx.clear();
  c->fp.uv = b->uv;
  c->fp.clone_idx = b->clone_idx;
  c->fp.n_meas = b->n_meas;
  c->fp.p_FinG = b->p_FinG;
  c->n_feats = b->n_feats;
  c->max_meas = b->max_meas;
  c->have_batch = true;
  c->pl_used_valid = false;
  c->range_lo = c->range_hi = -1;
  return 0;
}

// ---- the update step ---------------------------------------------------------------------------
// States above the tile factorization's limit (N > 288: e.g. 30 clones plus 50 landmarks).  A measurement batch never
// touches all of such a state: A = H^T H is non-zero on ns <= 288 involved columns s only.  With G = P[:, s]:
//     P+ = P - G (A - A Pss+ A) G^T ,   Pss+ = (Pss^-1 + A)^-1   (from (I + P A)^-1 = I - P+ A restricted to s),
// so the factorizations run on the ns x ns problem through the same tile kernels and the rest is three MFMA products.
static bool substate_ok(const ovp_ctx* c) { return c->n > OVP_TILECHOL_NMAX && c->sub_ns > 0 && c->sub_ns <= OVP_TILECHOL_NMAX; }

static int set_substate(ovp_ctx* c, const std::vector<int>& ids) {
  c->sub_ns = 0;
  if (c->n <= OVP_TILECHOL_NMAX || ids.empty() || (int)ids.size() > OVP_TILECHOL_NMAX) return 0;
  if (!c->sub_ids) HIPCHK(hipMalloc((void**)&c->sub_ids, sizeof(int) * (OVP_TILECHOL_NMAX + 16)));
  if (!c->sub_buf) HIPCHK(dalloc(&c->sub_buf, (size_t)6 * OVP_TILECHOL_NMAX * OVP_TILECHOL_NMAX));
  HIPCHK(hipMemcpyAsync(c->sub_ids, ids.data(), sizeof(int) * ids.size(), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));  // ids is the caller's temporary
  c->sub_ns = (int)ids.size();
  return 0;
}

static int chol_of_P(ovp_ctx* c, hipStream_t s) {
  const int n = c->n, ld = c->ld;
  if (c->pl_psd && n <= ovp_chol2_max_n() + 1) {
    // Positive SEMI-definite prior (plane loop, second attempt: state/StateHelper.cpp:159-187 never factors P, so the reference
    // updates such a covariance - right after StateHelper::clone the newest pose is an exact copy, :346-396).  ANY factor with
    // L0 L0^T = P serves the loop (P_k = L0 (I + L0^T A L0)^-1 L0^T is the matrix inversion lemma, no inverse of P in it): the
    // pivot-dropping Cholesky of the unit-diagonal form, L0 = D Lc with zero columns where P determines nothing.
    HIPCHK(ovp_launch_unit_diag(c->P, n, ld, c->W1, c->pl_crow, s));
    ovp::Chol2Job j;
    memset(&j, 0, sizeof(j));
    j.A = c->W1;
    j.n = n;
    j.ld = ld;
    j.mode = 0;
    j.flag = c->flags;
    j.piv_floor = 1e-12;  // regular pivots of the unit-diagonal form are >= 1 / cond (1e-8 at worst), dropped ones rounding noise
    j.Ldense = c->L;
    j.ldo = ld;
    HIPCHK(ovp_launch_chol2(&j, nullptr, nullptr, s));
    HIPCHK(ovp_launch_scale_rows(c->L, n, ld, c->pl_crow, s));
    return 0;
  }
  static const bool first_gen = getenv("OVP_TILECHOL_P") != nullptr;  // A/B: the first-generation kernel
  if (!first_gen && n <= ovp_chol2_max_n() + 1) {  // dense factor from the second-generation kernel
    ovp::Chol2Job j;
    memset(&j, 0, sizeof(j));
    j.A = c->P;
    j.n = n;
    j.ld = ld;
    j.mode = 0;
    j.flag = c->flags;
    j.Ldense = c->L;
    j.ldo = ld;
    return (int)ovp_launch_chol2(&j, nullptr, nullptr, s);
  }
  if (n <= OVP_TILECHOL_NMAX) return (int)ovp_launch_tilechol(c->P, c->L, nullptr, nullptr, n, ld, c->flags, 0, s);
  if (substate_ok(c)) return 0;  // the sub-state update factors Pss, not P
  return (int)ovp_launch_chol(c->P, c->L, n, ld, c->flags, 0, s);
}

// chol(T) behind an update: tile-packed factor + inverted diagonal blocks for k_fwdsub.  The second-generation kernel (k_chol2:
// fused elimination, role hand-over through LDS counters instead of workgroup barriers) took over from k_tilechol in round 3
// (OVP_TILECHOL_T=1 brings the first generation back for A/B runs).
static hipError_t chol_of_T(ovp_ctx* c, const double* T, int n, int ld, int add_identity, const int* cond, hipStream_t s) {
  static const bool first_gen = getenv("OVP_TILECHOL_T") != nullptr;
  if (first_gen || n > ovp_chol2_max_n() + 1)
    return ovp_launch_tilechol_unless(T, nullptr, c->Dinv, c->Ltp, n, ld, c->flags, add_identity, cond, s);
  return ovp_launch_chol2_packed(T, c->Dinv, c->Ltp, n, ld, c->flags, add_identity, cond, s);
}

static int ekf_substate(ovp_ctx* c, bool psd = false) {
  const int n = c->n, ld = c->ld, ns = c->sub_ns, lds = OVP_TILECHOL_NMAX;
  const size_t sz = (size_t)OVP_TILECHOL_NMAX * OVP_TILECHOL_NMAX;
  double *S_P = c->sub_buf, *S_A = S_P + sz, *S_L = S_A + sz, *S_W = S_L + sz, *S_T = S_W + sz, *S_Y = S_T + sz;
  hipStream_t s = c->stream;
  HIPCHK(ovp_launch_gather_block(c->P, ld, c->sub_ids, ns, S_P, lds, s));
  HIPCHK(ovp_launch_gather_block(c->Ab, ld, c->sub_ids, ns, S_A, lds, s));
  // Pss+ = Ls (I + Ls^T A Ls)^-1 Ls^T exactly as the full-state path does it
  if (psd) {
    // second attempt behind a failed chol(Pss): positive SEMI-definite prior (an exact stochastic clone) - any factor with
    // Ls Ls^T = Pss serves the identity above; the pivot-dropping Cholesky of the unit-diagonal form (cf. chol_of_P)
    HIPCHK(ovp_launch_unit_diag(S_P, ns, lds, S_W, c->dx, s));
    ovp::Chol2Job j;
    memset(&j, 0, sizeof(j));
    j.A = S_W;
    j.n = ns;
    j.ld = lds;
    j.mode = 0;
    j.flag = c->flags;
    j.piv_floor = 1e-12;
    j.Ldense = S_L;
    j.ldo = lds;
    HIPCHK(ovp_launch_chol2(&j, nullptr, nullptr, s));
    HIPCHK(ovp_launch_scale_rows(S_L, ns, lds, c->dx, s));
  } else {
    HIPCHK(ovp_launch_tilechol(S_P, S_L, nullptr, nullptr, ns, lds, c->flags, 0, s));
  }
  HIPCHK(ovp_launch_gemm4(0, 0, ns, ns, ns, S_A, lds, S_L, lds, S_W, lds, 0, 0, s));
  HIPCHK(ovp_launch_gemm4(1, 0, ns, ns, ns, S_L, lds, S_W, lds, S_T, lds, 1, 1, s));
  HIPCHK(ovp_launch_tilechol(S_T, nullptr, c->Dinv, c->Ltp, ns, lds, c->flags, 0, s));
  HIPCHK(ovp_launch_fwdsub(c->Ltp, c->Dinv, S_L, S_Y, ns, lds, 0, s));
  HIPCHK(ovp_launch_gemm4(1, 0, ns, ns, ns, S_Y, lds, S_Y, lds, S_P, lds, 0, 1, s));
  // Lambda = A - A Pss+ A
  HIPCHK(ovp_launch_gemm4(0, 0, ns, ns, ns, S_A, lds, S_P, lds, S_W, lds, 0, 0, s));
  HIPCHK(ovp_launch_gemm4(0, 0, ns, ns, ns, S_W, lds, S_A, lds, S_T, lds, 0, 1, s));
  HIPCHK(ovp_launch_mat_sub(S_A, S_T, S_T, ns, ns, lds, s));
  // P -= G Lambda G^T   (G in Y, G Lambda in W1, the product in T)
  HIPCHK(ovp_launch_gather_cols(c->P, ld, c->sub_ids, n, ns, c->Y, ld, s));
  HIPCHK(ovp_launch_gemm4(0, 0, n, ns, ns, c->Y, ld, S_T, lds, c->W1, ld, 0, 0, s));
  HIPCHK(ovp_launch_gemm4(0, 1, n, n, ns, c->W1, ld, c->Y, ld, c->T, ld, 0, 1, s));
  HIPCHK(ovp_launch_sub_sym_unless(c->P, c->T, n, ld, c->flags, s));  // (a failed factorization leaves the resident covariance alone)
  return 0;
}

static int ekf_from_gram(ovp_ctx* c, bool chol_p_done_on_stream2, bool publish = false) {
  const int n = c->n, ld = c->ld;
  if (!chol_p_done_on_stream2) {
    int rc = chol_of_P(c, c->stream);
    if (rc) return rc;
  } else {
    static const bool nojoin = getenv("OVP_DBG_NOJOIN") != nullptr;
    if (!nojoin && c->need_join) HIPCHK(hipStreamWaitEvent(c->stream, c->ev_join, 0));
  }
  const double* b = c->Ab + (size_t)n * ld;
  if (n <= OVP_TILECHOL_NMAX) {
    // W1 = A L ;  T = I + L^T W1 ;  Lt = chol(T) (+ inverses of its diagonal blocks).  L: chol(P), or the dense factor the plane
    // loop left (any M with M M^T = P gives P+ = M (I + M^T A M)^-1 M^T)
    const bool kept = c->use_kept_factor;
    const double* Lf = kept ? c->Lkeep : c->L;
    // leading block of T (reversed-order factor of P, see ovp_build_gate_gram_tail): n when the factor is the plain one
    const int nl = (!kept && chol_p_done_on_stream2 && c->point_nl > 0 && c->point_nl < n) ? c->point_nl : n;
    // diagonal amounts to take off at the end: CholJob::boost of the reversed-order factor (the first point_boost_n columns), or -
    // on the factor the plane loop left, which is a factor of P + diag(boost_vec) - the loop's own (all n entries, zero where none)
    const double* boost_ptr = kept ? (c->kept_boost ? c->boost_vec : nullptr) : c->boost;
    const int boost_n = kept ? (c->kept_boost ? n : 0) : (nl < n ? c->point_boost_n : 0);
    c->kept_boost = false;
    c->use_kept_factor = false;
    c->point_nl = 0;
    c->point_boost_n = 0;
    c->have_factor = false;  // P is about to change
    HIPCHK(ovp_launch_gemm4(0, 0, n, nl, n, c->Ab, ld, Lf, ld, c->W1, ld, 0, 0, c->stream));
    HIPCHK(ovp_launch_gemm4(1, 0, nl, nl, n, Lf, ld, c->W1, ld, c->T, ld, 1, 1, c->stream));
    HIPCHK(chol_of_T(c, c->T, nl, ld, 0, nullptr, c->stream));
    // V = Lt^-1 L^T ;  P+ = V^T V ;  dx = P+ b
    HIPCHK(ovp_launch_fwdsub_lead(c->Ltp, c->Dinv, Lf, c->Y, n, ld, kept ? 1 : (nl < n ? 2 : 0), nl, c->stream));
    // (skipped on the device when a factorization failed: the resident covariance then stays what it was, OVP_E_NOTSPD)
    HIPCHK(ovp_launch_gemm4c(1, 0, n, n, n, c->Y, ld, c->Y, ld, c->P, ld, 0, 1, c->flags, c->stream));
    if (publish) {
      // the last block of the dx kernel also publishes [flags | dx] to the pinned host block (no separate launch)
      const int words = (int)((16 + sizeof(double) * (size_t)n + 7) / 8);
      c->pub_seq = ++c->seq;
      HIPCHK(ovp_launch_dx_rows_boost(c->P, n, ld, b, c->dx, c->flags + 1, c->ticket, c->res_block, c->h_res_block_dev, words,
                                      (char*)c->h_res_block_dev + ((char*)c->h_seq - (char*)c->h_res_block), c->pub_seq, boost_ptr,
                                      boost_n, c->flags, c->stream));
      c->pub_pending = true;
    } else {
      HIPCHK(ovp_launch_dx_rows_boost(c->P, n, ld, b, c->dx, c->flags + 1, nullptr, nullptr, nullptr, 0, nullptr, 0u, boost_ptr, boost_n,
                                      c->flags, c->stream));
    }
    return 0;
  }
  if (substate_ok(c)) {
    int rc = ekf_substate(c);
    if (rc) return rc;
    if (publish) {
      const int words = (int)((16 + sizeof(double) * (size_t)n + 7) / 8);
      c->pub_seq = ++c->seq;
      HIPCHK(ovp_launch_dx_rows(c->P, n, ld, b, c->dx, c->flags + 1, c->ticket, c->res_block, c->h_res_block_dev, words,
                                (char*)c->h_res_block_dev + ((char*)c->h_seq - (char*)c->h_res_block), c->pub_seq, c->stream));
      c->pub_pending = true;
    } else {
      HIPCHK(ovp_launch_dx_rows(c->P, n, ld, b, c->dx, c->flags + 1, nullptr, nullptr, nullptr, 0, nullptr, 0u, c->stream));
    }
    return 0;
  }
  // large-state fallback when the measurements touch more than 288 columns: global-memory factorization
  HIPCHK(ovp_launch_gemm(0, 0, n, n, n, c->Ab, ld, c->L, ld, c->W1, ld, 0, c->stream));
  HIPCHK(ovp_launch_gemm(1, 0, n, n, n, c->L, ld, c->W1, ld, c->T, ld, 1, c->stream));
  HIPCHK(ovp_launch_chol(c->T, c->Lt, n, ld, c->flags, 0, c->stream));
  HIPCHK(ovp_launch_trsm_right_lt(c->L, c->Lt, c->Y, n, ld, c->stream));
  HIPCHK(ovp_launch_cov_finish(c->Y, n, ld, b, c->P, ld, c->dx, c->flags + 1, c->stream));
  return 0;
}

// Update of a covariance that is positive SEMI-definite (an exact stochastic clone before the next propagation, a zero-variance
// prior): P has no Cholesky factor, but the reference's own form needs none (state/StateHelper.cpp:159-187):
//   P+ = P - P H^T (H P H^T + I)^-1 H P,   with H := La^T, La La^T = A the (pivot-dropping) Cholesky factor of the information
// matrix of the batch - H^T H = A and H^T r = b is all the update depends on.  S = I + La^T P La is positive definite whatever P is.
// Runs after a failed chol(P) (flags[0]): the pair [A | b] is still in c->Ab, P was not touched (ovp_launch_gemm4c cancel flag).
static int ekf_sform(ovp_ctx* c) {
  const int n = c->n, ld = c->ld;
  hipStream_t s = c->stream;
  if (n > OVP_TILECHOL_NMAX || n > ovp_chol2_max_n()) {
    // above the tile factorization: the sub-state update once more, on the pivot-dropping factor of the involved block
    if (!substate_ok(c) || c->sub_ns > ovp_chol2_max_n()) return OVP_E_NOTSPD;
    HIPCHK(hipMemsetAsync(c->flags, 0, sizeof(int) * 4, s));
    const int rs = ekf_substate(c, true);
    if (rs) return rs;
    HIPCHK(ovp_launch_dx_rows(c->P, n, ld, c->Ab + (size_t)n * ld, c->dx, c->flags + 1, nullptr, nullptr, nullptr, 0, nullptr, 0u, s));
    HIPCHK(hipMemcpyAsync(c->h_dx, c->dx, sizeof(double) * n, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(c->h_flags, c->flags, sizeof(int) * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipMemsetAsync(c->flags, 0, sizeof(int) * 4, s));
    return 0;
  }
  HIPCHK(hipMemsetAsync(c->flags, 0, sizeof(int) * 4, s));
  HIPCHK(hipMemsetAsync(c->W1, 0, sizeof(double) * (size_t)n * ld, s));
  HIPCHK(ovp_launch_max_diag(c->Ab, n, ld, c->smallbuf, s));
  ovp::Chol2Job j;
  memset(&j, 0, sizeof(j));
  j.A = c->Ab;
  j.n = n;
  j.ld = ld;
  j.mode = 0;
  j.flag = c->flags + 3;
  j.piv_floor = 1e-13;  // directions that carry less than 1e-13 of the largest diagonal entry count as unobserved
  j.floor_scale = c->smallbuf;
  j.Ldense = c->W1;     // La (lower triangular, zero columns where a pivot was dropped)
  j.ldo = ld;
  HIPCHK(ovp_launch_chol2(&j, nullptr, nullptr, s));
  HIPCHK(ovp_launch_gemm4(0, 0, n, n, n, c->P, ld, c->W1, ld, c->Y, ld, 0, 0, s));   // Wm = P La
  HIPCHK(ovp_launch_gemm4(1, 0, n, n, n, c->W1, ld, c->Y, ld, c->T, ld, 1, 1, s));   // S = I + La^T Wm
  HIPCHK(ovp_launch_tilechol(c->T, nullptr, c->Dinv, c->Ltp, n, ld, c->flags, 0, s));
  HIPCHK(ovp_launch_fwdsub(c->Ltp, c->Dinv, c->Y, c->L, n, ld, 1, s));               // V = Ls^-1 Wm^T
  HIPCHK(ovp_launch_gemm4(1, 0, n, n, n, c->L, ld, c->L, ld, c->T, ld, 0, 1, s));    // V^T V
  HIPCHK(ovp_launch_sub_sym(c->P, c->T, n, ld, s));
  HIPCHK(ovp_launch_dx_rows(c->P, n, ld, c->Ab + (size_t)n * ld, c->dx, c->flags + 1, nullptr, nullptr, nullptr, 0, nullptr, 0u, s));
  HIPCHK(hipMemcpyAsync(c->h_dx, c->dx, sizeof(double) * n, hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(c->h_flags, c->flags, sizeof(int) * 4, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  HIPCHK(hipMemsetAsync(c->flags, 0, sizeof(int) * 4, s));
  return 0;
}

static int fill_feat_params(ovp_ctx* c, const ovp_update_opts* o);
static int ovp_build_gate_gram_tail(ovp_ctx* c, int n, int F);

extern "C" int ovp_msckf_build_gate_gram_async(ovp_ctx* c, const ovp_update_opts* o) {
  if (!c || !o) return OVP_E_ARG;
  if (!c->have_state || !c->have_cov || !c->have_batch) return OVP_E_STATE;
  if (c->fp.n_clones < 1) return OVP_E_STATE;
  const int n = c->n, F = c->n_feats;
  {
    int rc = fill_feat_params(c, o);
    if (rc) return rc;
  }
  // (the flag words were cleared by the previous ovp_msckf_fetch_results, or at creation)
  return ovp_build_gate_gram_tail(c, n, F);
}

static int fill_feat_params(ovp_ctx* c, const ovp_update_opts* o) {
  const int n = c->n, F = c->n_feats;
  ovp::FeatParams& fp = c->fp;
  fp.n_feats = F;
  fp.max_meas = c->max_meas;
  fp.do_fej = o->do_fej;
  fp.calmask = (o->do_calib_camera_pose ? 0x3Fu : 0u) | (o->do_calib_camera_intrinsics ? (0xFFu << 6) : 0u);
  for (int k = 0; k < 14; ++k) {
    fp.calcol[k] = (k < 6) ? c->calib_id + k : c->intr_id + (k - 6);
    if (!((fp.calmask >> k) & 1)) fp.calcol[k] = 0;
    else if (fp.calcol[k] < 0 || fp.calcol[k] >= n) return OVP_E_ARG;
  }
  fp.white_px = 1.0 / o->sigma_px;
  fp.chi2_mult = o->chi2_multiplier;
  fp.chi2_table = c->chi2_table;
  fp.P = c->P;
  fp.n = n;
  fp.ldp = c->ld;
  fp.G = c->G;
  fp.Bscr = c->Bscr;
  fp.ldg = c->ldg;
  fp.rec = c->rec;
  // per-feature results straight into the pinned host block (same layout as res_block): they cross PCIe while K1 runs
  fp.chi2 = (double*)((char*)c->h_res_block_dev + ((char*)c->chi2 - (char*)c->res_block));
  fp.accept = (unsigned char*)c->h_res_block_dev + ((char*)c->accept - (char*)c->res_block);
  fp.dbg_cycles = c->dbg_cycles;
  fp.slot = nullptr;
  fp.n_out = 0;
  fp.skip = nullptr;
  fp.range_lo = c->range_lo < 0 ? 0 : c->range_lo;
  fp.range_hi = c->range_lo < 0 ? 0x7fffffff : c->range_hi;
  if (o->skip_plane_used) {
    if (!c->pl_used_valid || !c->pl_used) return OVP_E_STATE;  // no plane update ran on this batch
    fp.skip = c->pl_used;
  }
  return 0;
}

static int build_gate_gram_tail_impl(ovp_ctx* c, int n, int F);
static int ovp_build_gate_gram_tail(ovp_ctx* c, int n, int F) {
  const int rc = build_gate_gram_tail_impl(c, n, F);
  if (rc && c->need_join) {
    // a failing exit behind the fork of chol(P): the main stream waits for the side stream before anything else is enqueued on it
    // (the regular join sits in ekf_from_gram, which a failed build never reaches)
    (void)hipEventRecord(c->ev_join, c->stream2);
    (void)hipStreamWaitEvent(c->stream, c->ev_join, 0);
    c->need_join = false;
  }
  return rc;
}
static int build_gate_gram_tail_impl(ovp_ctx* c, int n, int F) {
  ovp::FeatParams& fp = c->fp;
  // Round 5: the features that are not part of this update - consumed by an accepted plane (skip mask), outside this rank's index
  // range - no longer occupy rows of rec / G (they used to write 16 KB of zeros each, which K2 then read): the host knows both sets,
  // numbers the others in batch order and K1 reads its slot from host-mapped memory at the start of the feature wave; K2 runs on Fa
  // features.  Same sums over the same rows in the same order, the chunks / splits of K2 group fewer of them (OVP_NO_COMPACT=1: off).
  int Fa = F;
  {
    const bool ranged = c->range_lo >= 0;
    const bool masked = fp.skip != nullptr && c->pl_used_valid && (int)c->h_pl_used.size() == F;
    if ((ranged || masked) && F > 0 && !getenv("OVP_NO_COMPACT")) {
      const int lo = ranged ? c->range_lo : 0, hi = ranged ? (c->range_hi < F ? c->range_hi : F) : F;
      int k = 0;
      for (int f = 0; f < F; ++f) c->h_slot[f] = (f >= lo && f < hi && !(masked && c->h_pl_used[f])) ? k++ : -1;
      Fa = k;
      fp.slot = c->d_slot;
      fp.n_out = Fa;
    }
  }
  if (n > OVP_TILECHOL_NMAX) {  // the columns a point-feature batch can touch: clones, and calibration when it is estimated
    std::vector<int> ids;
    for (int cid : c->h_clone_id)
      for (int k = 0; k < 6; ++k) ids.push_back(cid + k);
    if (fp.calmask & 0x3Fu)
      for (int k = 0; k < 6; ++k) ids.push_back(c->calib_id + k);
    if (fp.calmask & (0xFFu << 6))
      for (int k = 0; k < 8; ++k) ids.push_back(c->intr_id + k);
    std::sort(ids.begin(), ids.end());
    ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
    int rs = set_substate(c, ids);
    if (rs) return rs;
  }
  // chol(P) does not depend on the measurements.  Default (mode 3): it rides in workgroup 0 of the fused feature kernel, on a
  // CU of its own, and is hidden behind the features; K2 follows on the main stream and nothing forks or joins.
  // Beside a one-wave-per-block K1 it must NOT run: K1 keeps every SIMD busy with two feature waves, and the CU that also
  // hosts the eight Cholesky waves finishes its feature blocks ~40 % later, which is the kernel's duration (K1 112 -> 157 us).
  // OVP_OVERLAP_MODE: 0 = no overlap, 1 = side stream beside K1, 2 = main stream after K1 with K2 beside it on the side
  // stream (the fallback when the fused kernel cannot take the batch), 3 (default) = fused.  (Also tried: chol(P) as its own
  // 160 KB-LDS launch on the side stream beside an 8-wave-workgroup K1 - the two launches did not overlap, K1 145 us.)
  // Round 5, mode 4 (default up to 1976 features): the features keep the fused kernel's shape - eight feature waves per workgroup,
  // each workgroup a CU of its own - but workgroup 0 factorizes nothing and returns at once; chol(P) runs as the second-generation
  // kernel (k_chol2, mode 0, reversed order / diagonal boost as CholJob has them) on the side stream, on a CU the features leave
  // free: 160 KB of LDS per feature workgroup keep the two off each other's SIMDs, and a round of 247 feature workgroups leaves a
  // CU on EVERY XCD (ovp_launch_feat_chol) - with 255 the side kernel's workgroup finds no CU on the XCD it is sent to and the
  // launches serialise (config 2, 2000 features: 316 against 274 us per update, measured), so above 1976 features mode 3 stays.
  // Closed-loop session (11 clones, ~100 features): msckf update 0.276 -> 0.260 ms per frame.
  const char* overlap_env = getenv("OVP_OVERLAP_MODE");  // (read per call: the tests switch it)
  int overlap_mode = overlap_env ? atoi(overlap_env) : 4;
  const bool fused_ok = c->n <= OVP_TILECHOL_NMAX && ovp_feat_chol_supported(&fp, c->n);
  if (overlap_mode == 4 && !(fused_ok && c->n <= ovp_chol2_max_n() && F <= ovp_feat_chol_side_capacity())) overlap_mode = 3;
  if (overlap_mode == 3 && !fused_ok) overlap_mode = 2;
  c->need_join = (overlap_mode == 1 || overlap_mode == 2);
  if (overlap_mode == 1) {
    HIPCHK(hipEventRecord(c->ev_fork, c->stream));
    HIPCHK(hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
    int rc = chol_of_P(c, c->stream2);
    if (rc) return rc;
    HIPCHK(hipEventRecord(c->ev_join, c->stream2));
  }
  // K1.  Events on the main stream are kept to a minimum (each one costs microseconds between dependent kernels): with
  // the kernel timer on, ev_k0 / ev_k1 bracket K1 and ev_k1 doubles as the fork point; otherwise one untimed fork event.
  if (c->ktimer) HIPCHK(hipEventRecord(c->ev_k0, c->stream));
  c->use_kept_factor = false;
  c->point_nl = 0;
  c->point_boost_n = 0;
  if (overlap_mode == 3 || overlap_mode == 4) {
    ovp::CholJob cj{c->P, c->L, nullptr, nullptr, c->n, c->ld, c->flags, 0, nullptr, 0, 0.0};
    if (c->have_factor && c->Lkeep) {  // the plane loop left M with M M^T = P: no chol(P) (cj.n = 0), the update runs on M
      cj.n = 0;
      c->use_kept_factor = true;
    } else {
      // The batch's information matrix lives on the clones and the estimated calibration.  When everything in front of the first
      // of those columns (the IMU block, dt) is untouched, the factor of P is taken in reversed index order: T = I + L^T A L is
      // then the identity outside its leading n - s0 columns, and both products, chol(T) and the substitution shrink with it
      // (config 2: 194 of 210 - 13 tile columns instead of 14) without a permutation of P.
      const bool no_flip = getenv("OVP_POINT_NO_FLIP") != nullptr;  // (read per call: the tests switch it)
      int s0 = c->n;
      for (int cid : c->h_clone_id) s0 = cid < s0 ? cid : s0;
      if (fp.calmask & 0x3Fu) s0 = c->calib_id < s0 ? c->calib_id : s0;
      if (fp.calmask & (0xFFu << 6)) s0 = c->intr_id < s0 ? c->intr_id : s0;
      if (!no_flip && s0 >= 8 && s0 < c->n) {
        cj.flip = 1;
        c->point_nl = c->n - s0;
        // the columns in front of the batch's take a relative diagonal boost inside the factorization that the end of the update
        // takes off again (CholJob::boost): exact, and an exact stochastic clone (IMU pose == newest clone) factors on this path
        const bool no_boost = getenv("OVP_POINT_NO_BOOST") != nullptr;  // (read per call: the tests switch it)
        if (!no_boost && s0 <= 64) {
          if (!c->boost) HIPCHK(dalloc(&c->boost, 64));
          cj.boost = c->boost;
          cj.boost_n = s0;
          cj.boost_rel = 1e-9;
          c->point_boost_n = s0;
        }
      }
    }
    if (overlap_mode == 4 && cj.n > 0) {
      ovp::Chol2Job j;
      memset(&j, 0, sizeof(j));
      j.A = c->P;
      j.n = c->n;
      j.ld = c->ld;
      j.mode = 0;
      j.flag = c->flags;
      j.Ldense = c->L;
      j.ldo = c->ld;
      j.flip = cj.flip;
      j.boost = cj.boost;
      j.boost_n = cj.boost_n;
      j.boost_rel = cj.boost_rel;
      HIPCHK(hipEventRecord(c->ev_fork, c->stream));
      c->need_join = true;  // (from here on the side stream may hold work: ovp_build_gate_gram_tail joins on a failing exit)
      HIPCHK(hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
      HIPCHK(ovp_launch_chol2(&j, nullptr, nullptr, c->stream2));
      HIPCHK(hipEventRecord(c->ev_join, c->stream2));
      cj.n = 0;
    }
    HIPCHK(ovp_launch_feat_chol(&fp, &cj, c->stream));
  } else {
    HIPCHK(ovp_launch_feat_gate(&fp, c->stream));
  }
  hipEvent_t fork_ev = c->ev_fork;
  if (c->ktimer) {
    fork_ev = c->ev_k1;
    c->kpending = true;
  }
  // K2 runs on the side stream in mode 2 (it is the shorter of the two branches: the join below then never stalls the
  // main stream, and the cross-queue wake-up latency sits at the START of the side branch, off the critical path)
  hipStream_t s2k = c->stream;
  if (overlap_mode == 2) {
    HIPCHK(hipEventRecord(fork_ev, c->stream));
    HIPCHK(hipStreamWaitEvent(c->stream2, fork_ev, 0));
    s2k = c->stream2;
    int rc = chol_of_P(c, c->stream);
    if (rc) return rc;
  } else {
    if (c->ktimer) HIPCHK(hipEventRecord(c->ev_k1, c->stream));
    if (overlap_mode == 0) {
      int rc = chol_of_P(c, c->stream);
      if (rc) return rc;
      HIPCHK(hipEventRecord(c->ev_join, c->stream));
    }
  }
  // K2 (on the Fa features that own rows of rec / G)
  const int used_chunks = Fa > 0 ? (2 * Fa + c->rows_per_chunk - 1) / c->rows_per_chunk : 0;
  if (Fa > 0) {
    int nsplit = (3 * Fa + 511) / 512;
    if (nsplit < 1) nsplit = 1;
    if (nsplit > c->n_split) nsplit = c->n_split;
    static const bool k2_split = getenv("OVP_K2_SPLIT") != nullptr;  // first version: two VALU / narrow-tile launches
    if (k2_split) {
      HIPCHK(ovp_launch_struct_gram(c->rec, fp.n_clones, Fa, c->rows_per_chunk, used_chunks, c->gramS, s2k));
      HIPCHK(ovp_launch_syrk(c->G, 3 * Fa, c->ldg, n + 1, nsplit, c->part, s2k));
    } else {
      HIPCHK(ovp_launch_gram_pair(c->rec, fp.n_clones, Fa, c->rows_per_chunk, used_chunks, c->gramS, c->G, 3 * Fa, c->ldg,
                                  n + 1, c->n_split, c->part, &nsplit, s2k));
    }
    HIPCHK(ovp_launch_reduce_gram(c->gramS, fp.n_clones, used_chunks, c->gramR, s2k));
    HIPCHK(ovp_launch_assemble(c->gramR, fp.n_clones, 1, c->part, nsplit, c->colmap, n, c->Ab, c->ld, s2k));
  } else {
    HIPCHK(hipMemsetAsync(c->Ab, 0, sizeof(double) * (size_t)(n + 1) * c->ld, s2k));
  }
  if (overlap_mode == 2) {
    // the Gram pair must be complete on the main stream when this call returns (the caller may all-reduce it there)
    HIPCHK(hipEventRecord(c->ev_join, c->stream2));
    static const bool nojoin = getenv("OVP_DBG_NOJOIN") != nullptr;  // timing experiment only (results are wrong)
    if (!nojoin) HIPCHK(hipStreamWaitEvent(c->stream, c->ev_join, 0));
  }
  return 0;
}

extern "C" int ovp_gram_buffer(ovp_ctx* c, double** Ab_dev, int* n_rows, int* ld) {
  if (!c || !Ab_dev) return OVP_E_ARG;
  *Ab_dev = c->Ab;
  if (n_rows) *n_rows = c->n + 1;
  if (ld) *ld = c->ld;
  return 0;
}

extern "C" int ovp_ekf_update_from_gram_async(ovp_ctx* c) {
  if (!c) return OVP_E_ARG;
  if (!c->have_cov) return OVP_E_STATE;
  int rc = ekf_from_gram(c, true, true);
  if (rc) return rc;
  if (c->ktimer) {
    HIPCHK(hipEventRecord(c->ev_t[3], c->stream));
    c->timed = true;
  }
  return 0;
}

extern "C" int ovp_msckf_fetch_results(ovp_ctx* c, double* dx_host, uint8_t* accepted_host, double* chi2_host,
                                       ovp_update_info* info) {
  if (!c) return OVP_E_ARG;
  const int n = c->n, F = c->n_feats;
  {
    // [flags | dx] are published by the last block of the dx kernel (or, on the fallback path, by a publish kernel);
    // chi2 / accept were written into the pinned block by K1 itself
    unsigned seq = c->pub_seq;
    if (!c->pub_pending) {
      const int words = (int)((16 + sizeof(double) * (size_t)n + 7) / 8);
      seq = ++c->seq;
      hipLaunchKernelGGL(k_publish_results, dim3(1), dim3(1024), 0, c->stream, (unsigned long long*)c->res_block,
                         (unsigned long long*)c->h_res_block_dev, words, (volatile unsigned*)((char*)c->h_res_block_dev +
                         ((char*)c->h_seq - (char*)c->h_res_block)), seq);
      HIPCHK(hipGetLastError());
    }
    c->pub_pending = false;
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (__atomic_load_n((const unsigned*)c->h_seq, __ATOMIC_ACQUIRE) != seq) {
      if ((++spins & 0xFFFu) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {
        HIPCHK(hipStreamSynchronize(c->stream));  // error path: surface a fault instead of spinning forever
        if (__atomic_load_n((const unsigned*)c->h_seq, __ATOMIC_ACQUIRE) != seq) return OVP_E_STATE;
        break;
      }
      __builtin_ia32_pause();
    }
  }
  if (c->h_flags[0]) {
    // chol(P) failed: the prior is only positive semi-definite.  Same update in the reference's S-form (no factor of P needed).
    int rs = ekf_sform(c);
    if (rs) return rs;
  }
  if (dx_host) memcpy(dx_host, c->h_dx, sizeof(double) * n);
  if (accepted_host && F) memcpy(accepted_host, c->h_accept, (size_t)F);
  if (chi2_host && F) memcpy(chi2_host, c->h_chi2, sizeof(double) * F);
  if (c->timed) {
    hipEventSynchronize(c->ev_t[3]);  // recorded behind the publishing kernel: may trail the sequence word by a moment
    // stage times while the kernel timer is on: [0] K1, [1] unused (K2 runs beside chol(P)), [2] chol(P) || K2 and the EKF
    // update, [3] total from the start of K1
    hipEventElapsedTime(&c->last_ms[0], c->ev_k0, c->ev_k1);
    c->last_ms[1] = 0.f;
    hipEventElapsedTime(&c->last_ms[2], c->ev_k1, c->ev_t[3]);
    hipEventElapsedTime(&c->last_ms[3], c->ev_k0, c->ev_t[3]);
    c->timed = false;
  }
  if (c->kpending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, c->ev_k0, c->ev_k1) == hipSuccess) {
      c->ktime_ms += ms;
      c->klaunches += 1;
    }
    c->kpending = false;
  }
  if (info) {
    memset(info, 0, sizeof(*info));
    // n_meas may live in caller-owned device memory (bind_device): read it back once per batch for the row count
    if (F && !c->h_nmeas_valid) {
      c->h_nmeas.resize(F);
      HIPCHK(hipMemcpy(c->h_nmeas.data(), c->fp.n_meas, sizeof(int) * F, hipMemcpyDeviceToHost));
      c->h_nmeas_valid = true;
    }
    for (int f = 0; f < F; ++f)
      if (c->h_accept[f]) {
        info->n_accepted++;
        info->n_rows += 2 * c->h_nmeas[f] - 3;
      }
    info->n_cols = 0;
    info->not_spd = c->h_flags[0];
    info->neg_diag = c->h_flags[1];
  }
  if (c->h_flags[0]) return OVP_E_NOTSPD;
  if (c->h_flags[1]) return OVP_E_NEGDIAG;
  return 0;
}

extern "C" int ovp_msckf_update(ovp_ctx* c, const ovp_update_opts* o, double* dx_host, uint8_t* accepted_host,
                                double* chi2_host, ovp_update_info* info) {
  const double t0 = host_now_ms();
  int rc = ovp_msckf_build_gate_gram_async(c, o);
  if (rc) return rc;
  rc = ovp_ekf_update_from_gram_async(c);
  if (rc) return rc;
  const double t1 = host_now_ms();
  rc = ovp_msckf_fetch_results(c, dx_host, accepted_host, chi2_host, info);
  c->host_acc[4] += t1 - t0;
  c->host_acc[5] += host_now_ms() - t1;
  c->host_acc[6] += 1.0;
  return rc;
}

// ---- feature-sharded point update over RCCL ---------------------------------------------------------------------
// One process per GPU; every rank holds the same covariance, pose tables and frame; a rank builds the information pair of its
// share of the point features, ONE ncclAllReduce(sum, f64) of [A | b] on the context's stream puts the summed pair on every rank,
// and every rank applies the identical update to its replica (no broadcast of P+).  RCCL is bound at first use with dlopen: a
// process that already carries an RCCL (torch ships one) shares it instead of loading a second copy, and the library itself keeps
// no link-time dependency on it.
namespace {
struct RcclApi {
  void* h = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, ovp_rccl_id, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool tried = false;
};
RcclApi g_rccl;
const int kNcclFloat64 = 8, kNcclUint8 = 1, kNcclSum = 0;  // ncclDataType_t / ncclRedOp_t of rccl.h (ncclFloat64 = ncclDouble = 8, ncclUint8 = 1, ncclSum = 0)

bool rccl_load() {
  if (g_rccl.tried) return g_rccl.AllReduce != nullptr;
  g_rccl.tried = true;
  const char* names[] = {getenv("OVP_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* nm : names) {
    if (!nm || !*nm) continue;
    void* h = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
    if (!h) continue;
    g_rccl.h = h;
    g_rccl.GetUniqueId = (int (*)(void*))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (int (*)(void**, int, ovp_rccl_id, int))dlsym(h, "ncclCommInitRank");
    g_rccl.CommDestroy = (int (*)(void*))dlsym(h, "ncclCommDestroy");
    g_rccl.AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(h, "ncclAllReduce");
    g_rccl.GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
    if (g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.AllReduce) return true;
    g_rccl = RcclApi();
    g_rccl.tried = true;
  }
  return false;
}
int rccl_rc(int r, const char* what) {
  if (r == 0) return 0;
  fprintf(stderr, "ovplane_hip: %s failed: %s\n", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
  return OVP_E_RCCL;
}
}  // namespace

extern "C" int ovp_rccl_unique_id(ovp_rccl_id* id) {
  if (!id) return OVP_E_ARG;
  if (!rccl_load()) return OVP_E_RCCL;
  return rccl_rc(g_rccl.GetUniqueId(id), "ncclGetUniqueId");
}

extern "C" int ovp_rccl_comm_create(const ovp_rccl_id* id, int rank, int world, int device, void** comm) {
  if (!id || !comm || world < 1 || rank < 0 || rank >= world) return OVP_E_ARG;
  if (!rccl_load()) return OVP_E_RCCL;
  HIPCHK(hipSetDevice(device));
  *comm = nullptr;
  return rccl_rc(g_rccl.CommInitRank(comm, world, *id, rank), "ncclCommInitRank");
}

extern "C" int ovp_rccl_comm_destroy(void* comm) {
  if (!comm) return OVP_E_ARG;
  if (!rccl_load()) return OVP_E_RCCL;
  return rccl_rc(g_rccl.CommDestroy(comm), "ncclCommDestroy");
}

// the collective alone, on the context's stream: for callers that drive the staged entry points themselves
extern "C" int ovp_rccl_allreduce_gram(ovp_ctx* c, void* nccl_comm) {
  if (!c || !nccl_comm) return OVP_E_ARG;
  if (!c->have_cov) return OVP_E_STATE;
  if (!rccl_load()) return OVP_E_RCCL;
  return rccl_rc(g_rccl.AllReduce(c->Ab, c->Ab, (size_t)(c->n + 1) * c->ld, kNcclFloat64, kNcclSum, nccl_comm, c->stream), "ncclAllReduce");
}

// this rank's balanced share of the features the update is about (the ones no accepted plane consumed when a mask is given):
// an index range of the resident batch - consecutive ranks tile it, consumed features inside are masked on the device.  Pure
// arithmetic (no context, no device): the CPU tests hold it against ov_plane_amd/dist.py: leftover_range at BASELINE config 4's size.
extern "C" int ovp_shard_range_of_mask(const uint8_t* used, int n_feats, int rank, int world, int* shard_lo, int* shard_hi) {
  if (!shard_lo || !shard_hi || n_feats < 0 || world < 1 || rank < 0 || rank >= world) return OVP_E_ARG;
  int nr = n_feats;
  if (used) {
    nr = 0;
    for (int f = 0; f < n_feats; ++f) nr += used[f] ? 0 : 1;
  }
  const int base = nr / world, rem = nr % world;
  const int a = rank * base + (rank < rem ? rank : rem), b = a + base + (rank < rem ? 1 : 0);
  *shard_lo = *shard_hi = 0;
  if (b <= a) return 0;
  if (!used) {
    *shard_lo = a;
    *shard_hi = b;
    return 0;
  }
  int k = 0;
  for (int f = 0; f < n_feats; ++f) {
    if (used[f]) continue;
    if (k == a) *shard_lo = f;
    if (k == b - 1) {
      *shard_hi = f + 1;
      break;
    }
    ++k;
  }
  return 0;
}

extern "C" int ovp_shard_range(ovp_ctx* c, const ovp_update_opts* o, int rank, int world, int* shard_lo, int* shard_hi) {
  if (!c || !o || !shard_lo || !shard_hi || world < 1 || rank < 0 || rank >= world) return OVP_E_ARG;
  if (!c->have_batch) return OVP_E_STATE;
  const int F = c->n_feats;
  const bool masked = o->skip_plane_used && c->pl_used_valid && (int)c->h_pl_used.size() == F;
  if (o->skip_plane_used && !masked) return OVP_E_STATE;  // no plane update ran on this batch
  return ovp_shard_range_of_mask(masked ? c->h_pl_used.data() : nullptr, F, rank, world, shard_lo, shard_hi);
}

// Errors of the sharded update are COLLECTIVE: a rank whose build failed still enters the all-reduce (with a zero pair) so that
// its peers do not wait for it forever, and says so in one more f64 word summed behind the pair; every rank then returns an error
// (its own, or OVP_E_PEER) from the same call.  k_peer_flag hands the summed word to the results block (flags[3]).
__global__ void k_peer_flag(const double* __restrict__ word, int* __restrict__ flag3) {
  if (*word != 0.0) *flag3 = 1;
}

extern "C" int ovp_msckf_update_sharded(ovp_ctx* c, const ovp_update_opts* o, void* nccl_comm, int rank, int world, double* dx_host,
                                        uint8_t* accepted_host, double* chi2_host, ovp_update_info* info, int* shard_lo,
                                        int* shard_hi) {
  // argument / state checks: identical on every rank of a correctly driven job (same frame, same options), taken before any
  // collective - a job whose ranks disagree HERE is mis-launched, not failing
  if (!c || !o || world < 1 || rank < 0 || rank >= world) return OVP_E_ARG;
  if (world > 1 && !nccl_comm) return OVP_E_ARG;
  if (!c->have_batch || !c->have_cov) return OVP_E_STATE;
  if (nccl_comm && !rccl_load()) return OVP_E_RCCL;
  int lo = 0, hi = 0;
  {
    const int rs = ovp_shard_range(c, o, rank, world, &lo, &hi);
    if (rs) return rs;
  }
  if (shard_lo) *shard_lo = lo;
  if (shard_hi) *shard_hi = hi;
  const double t0 = host_now_ms();
  const size_t pair_elems = (size_t)(c->n + 1) * c->ld;
  int rc = ovp_batch_set_range(c, lo, hi);
  if (!rc) rc = ovp_msckf_build_gate_gram_async(c, o);
  int rc_coll = 0;
  if (nccl_comm) {
    if (rc) {
      // rank-local failure (a HIP error in the build): a zero pair and a raised word, so that the peers' collective completes
      hipMemsetAsync(c->Ab, 0, sizeof(double) * pair_elems, c->stream);
      const double one = 1.0;
      hipMemcpyAsync(c->Ab + pair_elems, &one, sizeof(double), hipMemcpyHostToDevice, c->stream);
    } else {
      hipMemsetAsync(c->Ab + pair_elems, 0, sizeof(double), c->stream);
    }
    rc_coll = rccl_rc(g_rccl.AllReduce(c->Ab, c->Ab, pair_elems + 1, kNcclFloat64, kNcclSum, nccl_comm, c->stream), "ncclAllReduce");
    if (!rc && !rc_coll) {
      hipLaunchKernelGGL(k_peer_flag, dim3(1), dim3(1), 0, c->stream, (const double*)(c->Ab + pair_elems), c->flags + 3);
      rc = (int)hipGetLastError();
    }
  }
  if (!rc) rc = rc_coll;
  if (!rc) rc = ovp_ekf_update_from_gram_async(c);
  const double t1 = host_now_ms();
  if (!rc) {
    rc = ovp_msckf_fetch_results(c, dx_host, accepted_host, chi2_host, info);
    if (c->h_flags[3]) {
      // a peer's build failed: this rank's update ran on the pair of the healthy ranks only - every rank reports the failure and
      // the state is not to be used (the reference treats every failure on this path as fatal, state/StateHelper.cpp:185-187)
      c->have_cov = false;
      rc = OVP_E_PEER;
    }
  }
  c->host_acc[4] += t1 - t0;
  c->host_acc[5] += host_now_ms() - t1;
  c->host_acc[6] += 1.0;
  c->range_lo = c->range_hi = -1;
  return rc;
}

// Per-feature decisions of a sharded update on every rank: ovp_msckf_update_sharded fills accepted / chi2 for its own share only
// (zero elsewhere); a caller that erases rejected features from its feature vector on every replica (the Updater surface:
// update/UpdaterMSCKF.cpp:755-757) completes both arrays here - the shares are disjoint, so a sum is a gather.  Collective.
extern "C" int ovp_rccl_gather_decisions(ovp_ctx* c, void* nccl_comm, uint8_t* accepted_host, double* chi2_host) {
  if (!c || !accepted_host) return OVP_E_ARG;
  if (!c->have_batch) return OVP_E_STATE;
  const int F = c->n_feats;
  if (!nccl_comm || F == 0) return 0;
  if (!rccl_load()) return OVP_E_RCCL;
  // staged through the device result block (its chi2 / accept regions are unused while K1 writes straight to the pinned block)
  HIPCHK(hipMemcpyAsync(c->accept, accepted_host, (size_t)F, hipMemcpyHostToDevice, c->stream));
  if (chi2_host) HIPCHK(hipMemcpyAsync(c->chi2, chi2_host, sizeof(double) * F, hipMemcpyHostToDevice, c->stream));
  int rc = rccl_rc(g_rccl.AllReduce(c->accept, c->accept, (size_t)F, kNcclUint8, kNcclSum, nccl_comm, c->stream), "ncclAllReduce(accept)");
  if (!rc && chi2_host)
    rc = rccl_rc(g_rccl.AllReduce(c->chi2, c->chi2, (size_t)F, kNcclFloat64, kNcclSum, nccl_comm, c->stream), "ncclAllReduce(chi2)");
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(accepted_host, c->accept, (size_t)F, hipMemcpyDeviceToHost, c->stream));
  if (chi2_host) HIPCHK(hipMemcpyAsync(chi2_host, c->chi2, sizeof(double) * F, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

extern "C" int ovp_host_timing(ovp_ctx* c, int reset, double* out8) {
  if (!c) return OVP_E_ARG;
  if (out8) memcpy(out8, c->host_acc, sizeof(c->host_acc));
  if (reset) memset(c->host_acc, 0, sizeof(c->host_acc));
  return 0;
}

// device sequence shared by the plane update and the plane initialisation: feature kernel, Gram reduction, reduction to the
// state columns, range-energy factorisation, information-form update with the factor Mf (P = Mf Mf^T), gate.
// Leaves: V in c->Y, dx in c->dx, [chi2, ok, n_deg, pr] in c->pl_res + 4*pl, the extended Gram in c->pl_E.
static int plane_job_device(ovp_ctx* c, const ovp_update_opts* o, const ovp::FeatParams& fp, int pl, int start, int nf,
                            int in_state, int sid, double white_c, const double* Mf, int factor_dense, double thr,
                            int rows_live, int rows_u, int n_involved, int force = -1) {
  const int n = c->n, ld = c->ld, ldg = c->ldg;
  hipStream_t s = c->stream;
  ovp::PlaneParams pp;
  pp.feat_list = c->pl_featlist + start;
  pp.n_local = nf;
  pp.plane = pl;
  pp.in_state = in_state;
  pp.plane_sid = sid;
  pp.white_c = white_c;
  pp.cp = c->pl_cp;
  pp.cp_fej = c->pl_cp_fej;
  pp.cst = c->pl_cst;
  ovp::FeatParams fpl = fp;
  fpl.n = n;
  fpl.P = c->P;
  HIPCHK(ovp_launch_plane_feat(&fpl, &pp, nf, s));
  const int chunks = (2 * nf + c->rows_per_chunk - 1) / c->rows_per_chunk;
  HIPCHK(ovp_launch_struct_gram(c->rec, fp.n_clones, nf, c->rows_per_chunk, chunks, c->gramS, s));
  HIPCHK(ovp_launch_reduce_gram(c->gramS, fp.n_clones, chunks, c->gramR, s));
  int nsplit = (3 * nf + 511) / 512;
  if (nsplit < 1) nsplit = 1;
  if (nsplit > c->n_split) nsplit = c->n_split;
  HIPCHK(ovp_launch_syrk(c->G, 3 * nf, ldg, n + 4, nsplit, c->part, s));
  HIPCHK(ovp_launch_reduce_cst(c->pl_cst, nf, c->pl_cstsum, s));
  HIPCHK(ovp_launch_assemble_ext(c->gramR, fp.n_clones, c->part, nsplit, c->colmap, n, sid, c->pl_cstsum, c->pl_E, ldg, s));
  if (!in_state && c->pl_n_slam > 0)  // landmarks lying on this plane: one constraint row each (UpdaterMSCKF.cpp:545-552)
    HIPCHK(ovp_launch_plane_slam_rows(c->pl_E, ldg, n, pl + 1, c->pl_n_slam, c->pl_slam_i, c->pl_slam_i + c->pl_slam_cap,
                                      c->pl_slam_d, c->pl_slam_d + 3 * (size_t)c->pl_slam_cap, c->pl_cp + 3 * pl,
                                      c->pl_cp_fej + 3 * pl, white_c, fp.do_fej, c->pl_cstsum, s));
  HIPCHK(ovp_launch_plane_reduce_to_state(c->pl_E, ldg, n, in_state, c->Ab, ld, c->pl_cstsum + 9, c->pl_scal, s));
  // range part of the residual (regularised, diagonally normalised): its own Cholesky, independent of the update's -
  // side stream, joined before the gate (the two write different words of pl_scal)
  hipStream_t s2 = c->stream2;
  HIPCHK(hipEventRecord(c->ev_fork, s));
  HIPCHK(hipStreamWaitEvent(s2, c->ev_fork, 0));
  HIPCHK(ovp_launch_normalize_reg(c->Ab, ld, n, 1e-10, c->pl_An, c->pl_bn, s2));
  HIPCHK(ovp_launch_tilechol(c->pl_An, c->pl_Lr, c->pl_Dinv2, nullptr, n, ld, c->flags + 2, 0, s2));
  HIPCHK(ovp_launch_range_energy(c->pl_Lr, c->pl_Dinv2, c->pl_bn, n, ld, 1e-8, c->pl_scal, s2));
  HIPCHK(hipEventRecord(c->ev_join, s2));
  // EKF update in information form with the chained factor
  HIPCHK(ovp_launch_gemm4(0, 0, n, n, n, c->Ab, ld, Mf, ld, c->W1, ld, 0, 0, s));
  HIPCHK(ovp_launch_gemm4(1, 0, n, n, n, Mf, ld, c->W1, ld, c->T, ld, 1, 1, s));
  HIPCHK(chol_of_T(c, c->T, n, ld, 0, nullptr, s));  // (second-generation factorization where it fits, like every other chol(T))
  HIPCHK(ovp_launch_fwdsub(c->Ltp, c->Dinv, Mf, c->Y, n, ld, factor_dense, s));
  HIPCHK(ovp_launch_dx_from_factor(c->Y, n, ld, c->Ab + (size_t)n * ld, c->dx, c->pl_scal, s));
  HIPCHK(hipStreamWaitEvent(s, c->ev_join, 0));
  HIPCHK(ovp_launch_plane_gate(c->pl_scal, c->flags, thr, rows_live, rows_u, n_involved, force, c->pl_res + 4 * pl, s));
  return 0;
}

static int plane_buffers(ovp_ctx* c, int NP) {
  const int ld = c->ld;
  if (NP > c->pl_cap || !c->pl_E) {
    void* olds[] = {c->pl_sid, c->pl_cp, c->pl_cp_fej, c->pl_res, c->pl_dx};
    for (void* p : olds)
      if (p) hipFree(p);
    const int cap = NP + 8;
    HIPCHK(dalloc(&c->pl_sid, (size_t)cap));
    HIPCHK(dalloc(&c->pl_cp, (size_t)3 * cap));
    HIPCHK(dalloc(&c->pl_cp_fej, (size_t)3 * cap));
    HIPCHK(dalloc(&c->pl_res, (size_t)4 * cap));
    HIPCHK(dalloc(&c->pl_dx, (size_t)c->n_max * cap));
    c->pl_cap = cap;
    if (!c->pl_E) {
      const size_t ne = (size_t)(c->n_max + 4) * c->ldg;
      HIPCHK(dalloc(&c->pl_featlist, (size_t)c->f_max));
      HIPCHK(dalloc(&c->pl_cst, (size_t)c->f_max * 10));
      HIPCHK(dalloc(&c->pl_cstsum, 16));
      HIPCHK(dalloc(&c->pl_E, ne));
      HIPCHK(dalloc(&c->pl_An, (size_t)(c->n_max + 1) * ld));
      HIPCHK(dalloc(&c->pl_bn, (size_t)c->n_max));
      HIPCHK(dalloc(&c->pl_Lr, (size_t)(c->n_max + 1) * ld));
      HIPCHK(dalloc(&c->pl_Dinv2, (size_t)(ld / 16 + 1) * 256));
      HIPCHK(dalloc(&c->pl_scal, 8));
    }
  }
  return 0;
}

// ---- UpdaterMSCKF::update, per-plane loop ------------------------------------------------------
// First-generation plane loop (chained factor M <- M Lt^-T, one full EKF update per plane): kept for states above the
// register budget of k_chol2 (ovp_chol2_max_n() < n <= 288) and as a cross-check (OVP_PLANE_V1).
static int plane_update_v1(ovp_ctx* c, const ovp_update_opts* o, const ovp_plane_batch* pb, double* dx_planes,
                           uint8_t* plane_ok, double* plane_chi2, int* plane_dof, uint8_t* feat_used) {
  if (!c || !o || !pb || pb->n_planes < 0) return OVP_E_ARG;
  if (!c->have_state || !c->have_cov || !c->have_batch) return OVP_E_STATE;
  if (c->h_n_meas.empty() && c->n_feats > 0) return OVP_E_STATE;  // needs ovp_batch_upload (host copy of the layout)
  const int n = c->n, ld = c->ld, F = c->n_feats, NP = pb->n_planes, M = c->max_meas;
  if (n > OVP_TILECHOL_NMAX) return OVP_E_CAPACITY;
  if (feat_used) memset(feat_used, 0, (size_t)F);
  if (NP == 0) return 0;
  for (int k = 0; k < NP; ++k)
    if (pb->plane_state_id[k] >= 0 && pb->plane_state_id[k] + 3 > n) return OVP_E_ARG;
  const int n_slam = pb->n_slam > 0 ? pb->n_slam : 0;
  if (n_slam > 0 && (!pb->slam_plane || !pb->slam_state_id || !pb->slam_p || !pb->slam_p_fej)) return OVP_E_ARG;
  for (int q = 0; q < n_slam; ++q)
    if (pb->slam_state_id[q] < 0 || pb->slam_state_id[q] + 3 > n || pb->slam_plane[q] < 1 || pb->slam_plane[q] > NP) return OVP_E_ARG;
  int rc = fill_feat_params(c, o);
  if (rc) return rc;
  ovp::FeatParams fp = c->fp;
  // ---- host-side grouping (update/UpdaterMSCKF.cpp:204-229) ----
  struct PlaneJob { int pl, start, nf, rows_total, rows_live, rows_u, n_involved, in_state, sid; double thr; };
  std::vector<PlaneJob> jobs;
  std::vector<int> featlist;
  const int ncal = (o->do_calib_camera_pose ? 6 : 0) + (o->do_calib_camera_intrinsics ? 8 : 0);
  int max_nf = 1;
  for (int pl = 0; pl < NP; ++pl) {
    PlaneJob j;
    j.pl = pl;
    j.start = (int)featlist.size();
    j.nf = 0;
    j.rows_total = 0;
    j.rows_live = 0;
    j.sid = pb->plane_state_id[pl];
    j.in_state = j.sid >= 0;
    unsigned long long seen = 0ull;
    for (int f = 0; f < F; ++f) {
      if (pb->plane_of_feat[f] != pl + 1) continue;
      const int m = c->h_n_meas[f];
      if (m < 2) continue;
      if (m > OVP_MAX_MEAS_DEV) return OVP_E_CAPACITY;  // 2m bearing rows = one wavefront
      featlist.push_back(f);
      j.nf++;
      j.rows_total += 3 * m - 3;
      j.rows_live += 2 * m - 2;  // the m identical constraint rows are one direction (k_chol2 gate)
      for (int k = 0; k < m; ++k) seen |= 1ull << c->h_clone_idx[(size_t)f * M + k];
    }
    int ns_pl = 0;  // SLAM landmarks on this (out-of-state) plane: one row and three involved columns each
    if (!j.in_state)
      for (int q = 0; q < n_slam; ++q)
        if (pb->slam_plane[q] == pl + 1) ++ns_pl;
    if (j.nf == 0 || (!j.in_state && j.nf + ns_pl < 4)) {  // update/UpdaterMSCKF.cpp:316-317,384-396
      featlist.resize(j.start);
      continue;
    }
    j.rows_total += ns_pl;
    j.rows_live += ns_pl;
    const int c_ref = 6 * __builtin_popcountll(seen) + ncal + 3 * ns_pl;
    const int rows_c = j.rows_total > c_ref ? c_ref : j.rows_total;  // UpdaterPlane::measurement_compress_inplace
    j.rows_u = j.in_state ? rows_c : rows_c - 3;
    j.n_involved = c_ref + (j.in_state ? 3 : 0);
    if (!j.in_state) j.rows_total -= 3;
    if (!j.in_state) j.rows_live -= 3;
    if (j.rows_u < 1) {
      featlist.resize(j.start);
      continue;
    }
    j.thr = o->chi2_multiplier * ovp_chi2_quantile_095(j.rows_u);
    if (j.nf > max_nf) max_nf = j.nf;
    jobs.push_back(j);
  }
  rc = plane_buffers(c, NP);
  if (rc) return rc;
  hipStream_t s = c->stream;
  HIPCHK(hipMemsetAsync(c->pl_res, 0, sizeof(double) * 4 * NP, s));
  HIPCHK(hipMemsetAsync(c->pl_dx, 0, sizeof(double) * (size_t)n * NP, s));
  HIPCHK(hipMemcpyAsync(c->pl_sid, pb->plane_state_id, sizeof(int) * NP, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(c->pl_cp, pb->cp, sizeof(double) * 3 * NP, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(c->pl_cp_fej, pb->cp_fej, sizeof(double) * 3 * NP, hipMemcpyHostToDevice, s));
  if (!featlist.empty())
    HIPCHK(hipMemcpyAsync(c->pl_featlist, featlist.data(), sizeof(int) * featlist.size(), hipMemcpyHostToDevice, s));
  c->pl_n_slam = n_slam;
  if (n_slam > 0) {
    if (n_slam > c->pl_slam_cap) {
      if (c->pl_slam_i) hipFree(c->pl_slam_i);
      if (c->pl_slam_d) hipFree(c->pl_slam_d);
      c->pl_slam_cap = n_slam + 16;
      HIPCHK(hipMalloc((void**)&c->pl_slam_i, sizeof(int) * 2 * (size_t)c->pl_slam_cap));
      HIPCHK(dalloc(&c->pl_slam_d, (size_t)6 * c->pl_slam_cap));
    }
    HIPCHK(hipMemcpyAsync(c->pl_slam_i, pb->slam_plane, sizeof(int) * n_slam, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(c->pl_slam_i + c->pl_slam_cap, pb->slam_state_id, sizeof(int) * n_slam, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(c->pl_slam_d, pb->slam_p, sizeof(double) * 3 * n_slam, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(c->pl_slam_d + 3 * (size_t)c->pl_slam_cap, pb->slam_p_fej, sizeof(double) * 3 * n_slam, hipMemcpyHostToDevice, s));
  }
  HIPCHK(hipStreamSynchronize(s));  // the host vectors above go out of scope before the copies would otherwise run
  HIPCHK(hipMemsetAsync(c->flags, 0, sizeof(int) * 4, s));
  // factor of P, chained through the plane loop:  P = M M^T
  if (!jobs.empty()) {
    rc = chol_of_P(c, s);
    if (rc) return rc;
  }
  double* Mf = c->L;
  for (const PlaneJob& j : jobs) {
    rc = plane_job_device(c, o, fp, j.pl, j.start, j.nf, j.in_state, j.sid, 1.0 / o->sigma_constraint, Mf, 1, j.thr, j.rows_live,
                          j.rows_u, j.n_involved, pb->force_decision ? (int)pb->force_decision[j.pl] : -1);
    if (rc) return rc;
    HIPCHK(ovp_launch_plane_commit(c->pl_res + 4 * j.pl, c->Y, Mf, n, ld, c->dx, c->pl_dx + (size_t)j.pl * n, c->clone_R,
                                   c->clone_p, c->clone_id, fp.n_clones, c->cal, o->do_calib_camera_pose ? c->calib_id : -1,
                                   o->do_calib_camera_intrinsics ? c->intr_id : -1, c->pl_cp, c->pl_sid, NP, s));
    if (n_slam > 0)
      HIPCHK(ovp_launch_plane_commit_slam(c->pl_res + 4 * j.pl, c->dx, n_slam, c->pl_slam_i + c->pl_slam_cap, c->pl_slam_d, s));
  }
  c->pl_n_slam = 0;
  // P = M M^T
  if (!jobs.empty()) HIPCHK(ovp_launch_gemm4(0, 1, n, n, n, Mf, ld, Mf, ld, c->P, ld, 0, 1, s));
  // results
  std::vector<double> res(4 * (size_t)NP, 0.0);
  HIPCHK(hipMemcpyAsync(res.data(), c->pl_res, sizeof(double) * 4 * NP, hipMemcpyDeviceToHost, s));
  std::vector<double> dxh;
  if (dx_planes) HIPCHK(hipMemcpyAsync(dx_planes, c->pl_dx, sizeof(double) * (size_t)n * NP, hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(c->h_flags, c->flags, sizeof(int) * 4, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  for (int pl = 0; pl < NP; ++pl) {
    if (plane_ok) plane_ok[pl] = 0;
    if (plane_chi2) plane_chi2[pl] = 0.0;
    if (plane_dof) plane_dof[pl] = 0;
  }
  for (const PlaneJob& j : jobs) {
    const bool ok = res[4 * j.pl + 1] > 0.5;
    if (plane_ok) plane_ok[j.pl] = ok ? 1 : 0;
    if (plane_chi2) plane_chi2[j.pl] = res[4 * j.pl];
    if (plane_dof) plane_dof[j.pl] = j.rows_u;
    if (ok && feat_used)
      for (int k = 0; k < j.nf; ++k) feat_used[featlist[j.start + k]] = 1;
  }
  if (c->h_flags[0]) return OVP_E_NOTSPD;
  return 0;
}


// ---- UpdaterMSCKF::update, per-plane loop (second generation) ------------------------------------------------------------
// See k_plane2.hip for the algebra.  Everything of a call is enqueued without a host synchronisation: the per-call tables go
// through one pinned staging block, the results come back through one pinned block read after a single stream sync.
struct PlaneJobH { int pl, start, nf, rows_total, rows_live, rows_u, n_involved, in_state, sid, n_inv_cols, ns_pl; double thr; };

extern "C" int ovp_msckf_plane_update(ovp_ctx* c, const ovp_update_opts* o, const ovp_plane_batch* pb, double* dx_planes,
                                      uint8_t* plane_ok, double* plane_chi2, int* plane_dof, uint8_t* feat_used);

static int ensure_pl_used(ovp_ctx* c) {
  if (!c->pl_used) HIPCHK(hipMalloc((void**)&c->pl_used, (size_t)c->f_max + 16));
  return 0;
}

static int plane2_buffers(ovp_ctx* c, int NP, size_t stage_bytes, size_t res_bytes) {
  const int ld = c->ld;
  if (!c->pl_Tbuf) {
    const size_t nn = (size_t)(c->n_max + 1) * ld;
    HIPCHK(dalloc(&c->pl_Tbuf, 2 * nn));
    HIPCHK(dalloc(&c->pl_crow, (size_t)c->n_max + 16));
    HIPCHK(dalloc(&c->pl_dxlast, (size_t)c->n_max + 16));
    HIPCHK(hipMalloc((void**)&c->pl_cur, 16));
    HIPCHK(hipMalloc((void**)&c->pl_range_done, 16));
    HIPCHK(hipMemset(c->pl_range_done, 0, 16));
    HIPCHK(dalloc(&c->pl_xbuf, (size_t)9 * 18 * 256));
    HIPCHK(dalloc(&c->pl_xy, (size_t)c->n_max + 32));
    HIPCHK(hipMalloc((void**)&c->pl_xflag, sizeof(unsigned) * 64));
    HIPCHK(hipMemset(c->pl_xflag, 0, sizeof(unsigned) * 64));
  }
  if (NP > c->pl2_cap) {
    if (c->pl_perm) hipFree(c->pl_perm);
    c->pl2_cap = NP + 8;
    HIPCHK(hipMalloc((void**)&c->pl_perm, sizeof(int) * (size_t)c->pl2_cap * c->n_max));
  }
  if (stage_bytes > c->pl_stage_cap) {
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->pl_hstage) hipHostFree(c->pl_hstage);
    if (c->pl_dstage) hipFree(c->pl_dstage);
    c->pl_stage_cap = stage_bytes + 4096;
    HIPCHK(hipHostMalloc(&c->pl_hstage, c->pl_stage_cap, hipHostMallocDefault));
    HIPCHK(hipMalloc(&c->pl_dstage, c->pl_stage_cap));
  }
  if (res_bytes > c->pl_hres_cap) {
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->pl_hres) hipHostFree(c->pl_hres);
    c->pl_hres_cap = res_bytes + 4096;
    HIPCHK(hipHostMalloc(&c->pl_hres, c->pl_hres_cap, hipHostMallocDefault));
  }
  return 0;
}

// ---- a selection of state columns as a state of its own (plane loop in its own order, plane initialisation on the marginal) ----
// Device block [ids | inverse | clone ids | column map] of the selection `ids` (pos = its inverse, -1 = not selected), staged in the
// context's pinned block and sent on stream s; the kernels then address the selection through these tables instead of the state's.
struct SubTables {
  const int* d_ids = nullptr;
  const int* d_inv = nullptr;
  int* d_clone_id = nullptr;
  ovp::ColMap* d_colmap = nullptr;
  int calib_sub = -1, intr_sub = -1;
  std::vector<int> clone_sub;
};
static int sub_tables_upload(ovp_ctx* c, const ovp_update_opts* o, const std::vector<int>& ids, const std::vector<int>& pos,
                             SubTables* t, hipStream_t s) {
  const int n = c->n, C = c->fp.n_clones, ns = (int)ids.size();
  const size_t o_ids = 0, o_inv = sizeof(int) * (size_t)(c->n_max + 16), o_tab = 2 * o_inv;
  const size_t tab_bytes = sizeof(int) * (size_t)(c->c_max + 16) + sizeof(ovp::ColMap) * (size_t)c->n_max;
  const size_t blk_bytes = o_tab + tab_bytes;
  if (!c->pl_sub_tab) {
    HIPCHK(hipMalloc(&c->pl_sub_tab, blk_bytes));
    HIPCHK(hipHostMalloc(&c->pl_sub_htab, blk_bytes, hipHostMallocDefault));
  }
  if (c->ev_subtab) HIPCHK(hipEventSynchronize(c->ev_subtab));  // the pinned block fed the copy of the previous call (long done)
  else HIPCHK(hipEventCreateWithFlags(&c->ev_subtab, hipEventDisableTiming));
  char* hb = (char*)c->pl_sub_htab;
  memset(hb, 0, blk_bytes);
  int* h_ids = (int*)(hb + o_ids);
  int* h_inv = (int*)(hb + o_inv);
  int* t_clone = (int*)(hb + o_tab);
  ovp::ColMap* t_cm = (ovp::ColMap*)(hb + o_tab + sizeof(int) * (size_t)(c->c_max + 16));
  memcpy(h_ids, ids.data(), sizeof(int) * (size_t)ns);
  for (int col = 0; col < n; ++col) h_inv[col] = pos[col] >= 0 ? pos[col] : 0;
  t->clone_sub.assign((size_t)C, 0);
  for (int i = 0; i < C; ++i) {
    t->clone_sub[i] = t_clone[i] = pos[c->h_clone_id[i]];
    for (int k = 0; k < 6; ++k) {
      ovp::ColMap& m = t_cm[t->clone_sub[i] + k];
      m.kind = 1;
      m.idx = i;
      m.off = k;
    }
  }
  t->calib_sub = (c->calib_id >= 0 && c->calib_id + 6 <= n && pos[c->calib_id] >= 0) ? pos[c->calib_id] : -1;
  t->intr_sub = (c->intr_id >= 0 && c->intr_id + 8 <= n && pos[c->intr_id] >= 0) ? pos[c->intr_id] : -1;
  if (t->calib_sub >= 0 && o->do_calib_camera_pose)
    for (int k = 0; k < 6; ++k) {
      t_cm[t->calib_sub + k].kind = 2;
      t_cm[t->calib_sub + k].idx = k;
    }
  if (t->intr_sub >= 0 && o->do_calib_camera_intrinsics)
    for (int k = 0; k < 8; ++k) {
      t_cm[t->intr_sub + k].kind = 2;
      t_cm[t->intr_sub + k].idx = 6 + k;
    }
  HIPCHK(hipMemcpyAsync(c->pl_sub_tab, hb, blk_bytes, hipMemcpyHostToDevice, s));
  HIPCHK(hipEventRecord(c->ev_subtab, s));
  t->d_ids = (const int*)((char*)c->pl_sub_tab + o_ids);
  t->d_inv = (const int*)((char*)c->pl_sub_tab + o_inv);
  t->d_clone_id = (int*)((char*)c->pl_sub_tab + o_tab);
  t->d_colmap = (ovp::ColMap*)((char*)c->pl_sub_tab + o_tab + sizeof(int) * (size_t)(c->c_max + 16));
  return 0;
}
// the context's view of the state while a selection stands in for it, and back
struct SubSaved {
  int n, calib_id, intr_id;
  double* P;
  int* clone_id;
  ovp::ColMap* colmap;
  const int* fp_clone_id;
  std::vector<int> h_clone_id;
};
static SubSaved sub_enter(ovp_ctx* c, const SubTables& t, int ns, double* Psub) {
  SubSaved sv{c->n, c->calib_id, c->intr_id, c->P, c->clone_id, c->colmap, c->fp.clone_id, c->h_clone_id};
  c->n = ns;
  c->P = Psub;
  c->calib_id = t.calib_sub;
  c->intr_id = t.intr_sub;
  c->clone_id = t.d_clone_id;
  c->fp.clone_id = c->clone_id;
  c->colmap = t.d_colmap;
  c->h_clone_id = t.clone_sub;
  return sv;
}
static void sub_leave(ovp_ctx* c, const SubSaved& sv) {
  c->n = sv.n;
  c->P = sv.P;
  c->calib_id = sv.calib_id;
  c->intr_id = sv.intr_id;
  c->clone_id = sv.clone_id;
  c->fp.clone_id = sv.fp_clone_id;
  c->colmap = sv.colmap;
  c->h_clone_id = sv.h_clone_id;
}

// ---- the plane loop in the loop's own column order (update/UpdaterMSCKF.cpp:413-649 has no size limit) --------------------------
// A plane's rows touch the clones, the calibration, its own closest point when it is a state variable and the SLAM landmarks lying
// on it (out-of-state planes).  Two things follow:
//  (1) LEADING BLOCK.  With P0 = L0 L0^T in the order [clones + calibration | the planes' own columns in processing order |
//      everything no plane of the call involves (IMU, dt, other landmarks)], A_k is zero outside the columns involved so far and L0
//      is lower triangular, so L0^T A_k L0 is zero outside that LEADING block: T_k = blockdiag(T_lead, I).  Plane k's products and
//      its factorization run on nl_k = 6 C + calibration + (own columns of the planes up to k) columns instead of n (config 3:
//      194 .. 224 of 240 - 13 to 14 tile steps of k_chol2 instead of 15, and shorter ones); only dx = L0[:, 0:nl] y and the commit
//      see all n rows.  The covariance is permuted once in front of the loop and once behind it.
//  (2) SUB-STATE.  Above the factorization's limit (n > 287) the loop runs on the marginal P0[s, s] of the involved columns s
//      (ns <= 287; same order) - same kernels, the state tables addressed through remapped column ids - and the rest of the state
//      follows from the push-through identity (k_plane_sub_accum for dx, the point path's  P -= G (A - A Pss+ A) G^T  for P).
static int plane_update_ordered(ovp_ctx* c, const ovp_update_opts* o, const ovp_plane_batch* pb, double* dx_planes,
                                uint8_t* plane_ok, double* plane_chi2, int* plane_dof, uint8_t* feat_used) {
  const double t_entry = host_now_ms();
  const int n = c->n, ld = c->ld, NP = pb->n_planes, C = c->fp.n_clones;
  const int n_slam = pb->n_slam > 0 ? pb->n_slam : 0;
  if (n_slam > 0 && (!pb->slam_plane || !pb->slam_state_id || !pb->slam_p || !pb->slam_p_fej)) return OVP_E_ARG;
  // ---- column order: first involvement ----
  std::vector<int> ids, pos((size_t)n, -1);
  ids.reserve((size_t)n);
  bool bad_id = false;
  auto place = [&](int id, int sz) {
    if (id < 0 || id + sz > n) {
      bad_id = true;
      return;
    }
    for (int k = 0; k < sz; ++k)
      if (pos[id + k] < 0) {
        pos[id + k] = (int)ids.size();
        ids.push_back(id + k);
      }
  };
  for (int i = 0; i < C; ++i) place(c->h_clone_id[i], 6);
  if (o->do_calib_camera_pose) place(c->calib_id, 6);
  if (o->do_calib_camera_intrinsics) place(c->intr_id, 8);
  for (int q = 0; q < n_slam; ++q)
    if (pb->slam_plane[q] < 1 || pb->slam_plane[q] > NP || pb->slam_state_id[q] < 0 || pb->slam_state_id[q] + 3 > n) return OVP_E_ARG;
  c->pl_nl.assign((size_t)(NP > 0 ? NP : 1), 0);
  for (int k = 0; k < NP; ++k) {
    if (pb->plane_state_id[k] >= 0) place(pb->plane_state_id[k], 3);
    else
      for (int q = 0; q < n_slam; ++q)
        if (pb->slam_plane[q] == k + 1) place(pb->slam_state_id[q], 3);
    c->pl_nl[k] = (int)ids.size();
  }
  if (bad_id) return OVP_E_ARG;
  const int n_inv = (int)ids.size();
  const bool full = n <= ovp_chol2_max_n();  // the whole state fits one factorization: the rest rides along behind the leading block
  if (!full && n_inv > ovp_chol2_max_n()) return OVP_E_CAPACITY;  // the planes of this call involve more columns than one factorization holds
  if (full)
    for (int col = 0; col < n; ++col)
      if (pos[col] < 0) {
        pos[col] = (int)ids.size();
        ids.push_back(col);
      }
  const int ns = (int)ids.size();
  hipStream_t s = c->stream;
  if (!full) {
    if (!c->pl_Asum) HIPCHK(dalloc(&c->pl_Asum, (size_t)c->n_max * ld));
    if (NP > c->pl_U_cap) {
      if (c->pl_U) hipFree(c->pl_U);
      c->pl_U_cap = NP + 8;
      HIPCHK(dalloc(&c->pl_U, (size_t)c->pl_U_cap * ld));
    }
  }
  std::vector<int> sid_sub(NP > 0 ? NP : 1, -1), slam_sub(n_slam > 0 ? n_slam : 1, 0);
  for (int k = 0; k < NP; ++k) sid_sub[k] = pb->plane_state_id[k] >= 0 ? pos[pb->plane_state_id[k]] : -1;
  for (int q = 0; q < n_slam; ++q) {
    // a landmark listed on a plane that IS in the state takes no part in the loop (UpdaterMSCKF.cpp:240-241): park it on column 0
    // (in the full order every column has a position; on a marginal the landmark's columns may be absent)
    const int p0 = (pb->slam_state_id[q] >= 0 && pb->slam_state_id[q] + 3 <= n) ? pos[pb->slam_state_id[q]] : -1;
    if (p0 < 0 && pb->plane_state_id[pb->slam_plane[q] - 1] < 0) return OVP_E_ARG;
    slam_sub[q] = p0 >= 0 ? p0 : 0;
  }
  if (c->pl_ktimer) {
    while (c->pl_ev_loop.size() < 2) {
      hipEvent_t e;
      HIPCHK(hipEventCreate(&e));
      c->pl_ev_loop.push_back(e);
    }
    HIPCHK(hipEventRecord(c->pl_ev_loop[0], s));
  }
  // ---- remapped tables: [ids | inverse | clone ids | column map] ----
  SubTables st;
  {
    const int rt = sub_tables_upload(c, o, ids, pos, &st, s);
    if (rt) return rt;
  }
  const int* d_ids = st.d_ids;
  const int* d_inv = st.d_inv;
  // full order: the columns behind the involved ones take a diagonal boost that the un-permutation behind the loop takes off
  // again (k_gather_block_boost) - an exact stochastic clone then factors at the first attempt
  const bool no_boost = getenv("OVP_PL_NO_BOOST") != nullptr;  // (read per call)
  c->pl_boost_active = full && n_inv < ns && !no_boost;
  if (c->pl_boost_active) {
    if (!c->boost_vec) HIPCHK(dalloc(&c->boost_vec, (size_t)c->n_max + 16));
    HIPCHK(ovp_launch_gather_block_boost(c->P, ld, d_ids, ns, c->P_tmp, ld, n_inv, 1e-9, c->boost_vec, s));
  } else {
    HIPCHK(ovp_launch_gather_block(c->P, ld, d_ids, ns, c->P_tmp, ld, s));
  }
  if (!full) {
    HIPCHK(hipMemsetAsync(c->pl_Asum, 0, sizeof(double) * (size_t)ns * ld, s));
    HIPCHK(hipMemsetAsync(c->pl_U, 0, sizeof(double) * (size_t)NP * ld, s));
  }
  // ---- the loop in the new order ----
  ovp_plane_batch pbs = *pb;
  pbs.plane_state_id = sid_sub.data();
  pbs.slam_state_id = slam_sub.data();
  std::vector<double> dx_sub((size_t)ns * (NP > 0 ? NP : 1), 0.0);
  SubSaved sv = sub_enter(c, st, ns, c->P_tmp);
  c->pl_sub_active = true;
  c->pl_sub_rest = !full;
  c->pl_scatter_dst = full ? sv.P : nullptr;  // full order: the loop's covariance product is un-permuted straight into the resident P
  c->pl_scatter_ids = d_inv;
  c->pl_t_entry = t_entry;
  const int rc = ovp_msckf_plane_update(c, o, &pbs, dx_sub.data(), plane_ok, plane_chi2, plane_dof, feat_used);
  c->pl_sub_active = false;
  c->pl_sub_rest = false;
  c->pl_scatter_dst = nullptr;
  double* Pss_new = c->P;  // = P_tmp: the marginal after the loop
  sub_leave(c, sv);
  if (rc) return rc;  // the resident covariance was not touched (the device tables may have been: a loop that fails after
                      // accepting planes has marked them invalid, have_state = false - INTEGRATION.md section 5)
  if (full) {
    if (dx_planes)
      for (int k = 0; k < NP; ++k)
        for (int i = 0; i < ns; ++i) dx_planes[(size_t)k * n + ids[i]] = dx_sub[(size_t)k * ns + i];
    return 0;
  }
  // ---- the rest of the state ----
  // Lambda = Asum - Asum Pss+ Asum ;  P -= G Lambda G^T ;  dx_k = G u_k     (G = P0[:, s] in Y)
  HIPCHK(ovp_launch_gemm4(0, 0, ns, ns, ns, c->pl_Asum, ld, Pss_new, ld, c->W1, ld, 0, 0, s));
  HIPCHK(ovp_launch_gemm4(0, 0, ns, ns, ns, c->W1, ld, c->pl_Asum, ld, c->T, ld, 0, 1, s));
  HIPCHK(ovp_launch_mat_sub(c->pl_Asum, c->T, c->T, ns, ns, ld, s));
  HIPCHK(ovp_launch_gather_cols(c->P, ld, d_ids, n, ns, c->Y, ld, s));
  if (dx_planes && NP > 0) {
    // rows = planes: DX (NP x n) = U (NP x ns) G^T
    HIPCHK(ovp_launch_gemm4(0, 1, NP, n, ns, c->pl_U, ld, c->Y, ld, c->Lt, ld, 0, 0, s));
    HIPCHK(hipMemcpy2DAsync(dx_planes, sizeof(double) * n, c->Lt, sizeof(double) * ld, sizeof(double) * n, NP, hipMemcpyDeviceToHost, s));
  }
  HIPCHK(ovp_launch_gemm4(0, 0, n, ns, ns, c->Y, ld, c->T, ld, c->W1, ld, 0, 0, s));
  HIPCHK(ovp_launch_gemm4(0, 1, n, n, ns, c->W1, ld, c->Y, ld, c->L, ld, 0, 1, s));
  HIPCHK(ovp_launch_sub_sym(c->P, c->L, n, ld, s));
  HIPCHK(hipStreamSynchronize(s));
  if (dx_planes)  // the involved entries straight from the loop (the product above agrees with them to rounding)
    for (int k = 0; k < NP; ++k)
      for (int i = 0; i < ns; ++i) dx_planes[(size_t)k * n + ids[i]] = dx_sub[(size_t)k * ns + i];
  return 0;
}

extern "C" int ovp_msckf_plane_update(ovp_ctx* c, const ovp_update_opts* o, const ovp_plane_batch* pb, double* dx_planes,
                                      uint8_t* plane_ok, double* plane_chi2, int* plane_dof, uint8_t* feat_used) {
  if (!c || !o || !pb || pb->n_planes < 0) return OVP_E_ARG;
  if (!c->have_state || !c->have_cov || !c->have_batch) return OVP_E_STATE;
  if (c->h_n_meas.empty() && c->n_feats > 0) return OVP_E_STATE;  // needs ovp_batch_upload (host copy of the layout)
  const double t_entry = c->pl_sub_active ? c->pl_t_entry : host_now_ms();
  c->have_factor = false;
  const int n = c->n, ld = c->ld, F = c->n_feats, NP = pb->n_planes, M = c->max_meas;
  // skip_plane_used is an option of the POINT update that follows; the plane loop itself produces the mask
  ovp_update_opts o_local = *o;
  o_local.skip_plane_used = 0;
  o = &o_local;
  c->pl_used_valid = false;
  int rcu = ensure_pl_used(c);
  if (rcu) return rcu;
  static const bool force_v1 = getenv("OVP_PLANE_V1") != nullptr;
  const bool natural_order = getenv("OVP_PL_NATURAL_ORDER") != nullptr;  // A/B: the loop on all n columns in the state's order
  if (!force_v1 && !c->pl_sub_active && NP > 0 && (n > ovp_chol2_max_n() || !natural_order))
    return plane_update_ordered(c, o, pb, dx_planes, plane_ok, plane_chi2, plane_dof, feat_used);
  if (n > ovp_chol2_max_n() || force_v1) {
    // first generation (no device-side mask of its own): the host mask it reports is mirrored into pl_used
    std::vector<uint8_t> used_h((size_t)(F > 0 ? F : 1), 0);
    const int rc1 = plane_update_v1(c, o, pb, dx_planes, plane_ok, plane_chi2, plane_dof, used_h.data());
    if (rc1) return rc1;
    if (F) HIPCHK(hipMemcpy(c->pl_used, used_h.data(), (size_t)F, hipMemcpyHostToDevice));
    if (feat_used && F) memcpy(feat_used, used_h.data(), (size_t)F);
    c->h_pl_used.assign(used_h.begin(), used_h.begin() + F);
    c->pl_used_valid = true;
    return 0;
  }
  if (feat_used) memset(feat_used, 0, (size_t)F);
  for (int pl = 0; pl < NP; ++pl) {
    if (plane_ok) plane_ok[pl] = 0;
    if (plane_chi2) plane_chi2[pl] = 0.0;
    if (plane_dof) plane_dof[pl] = 0;
  }
  if (dx_planes && NP > 0) memset(dx_planes, 0, sizeof(double) * (size_t)n * NP);
  if (NP == 0) {  // a frame without planes: nothing is consumed, and a point update with skip_plane_used may follow
    if (F) HIPCHK(hipMemsetAsync(c->pl_used, 0, (size_t)F, c->stream));
    c->h_pl_used.assign((size_t)F, 0);
    c->pl_used_valid = true;
    return 0;
  }
  for (int k = 0; k < NP; ++k)
    if (pb->plane_state_id[k] >= 0 && pb->plane_state_id[k] + 3 > n) return OVP_E_ARG;
  const int n_slam = pb->n_slam > 0 ? pb->n_slam : 0;
  if (n_slam > 0 && (!pb->slam_plane || !pb->slam_state_id || !pb->slam_p || !pb->slam_p_fej)) return OVP_E_ARG;
  for (int q = 0; q < n_slam; ++q)
    if (pb->slam_state_id[q] < 0 || pb->slam_state_id[q] + 3 > n || pb->slam_plane[q] < 1 || pb->slam_plane[q] > NP) return OVP_E_ARG;
  int rc = fill_feat_params(c, o);
  if (rc) return rc;
  ovp::FeatParams fp = c->fp;
  fp.skip = nullptr;
  fp.range_lo = 0;  // the plane loop always walks the whole batch
  fp.range_hi = 0x7fffffff;
  // ---- what does not depend on the grouping goes to the device first: the fills and chol(P) (~70 us) run while the host sorts the
  // features by plane and builds the per-plane tables (~40 us at config 3, during which the stream used to be idle) ----
  const size_t res_bytes = sizeof(double) * (4 * (size_t)NP + (size_t)n * NP) + (size_t)F + 64;
  rc = plane_buffers(c, NP);  // shared with the first generation: pl_res, pl_dx, pl_cst, pl_An, ...
  if (rc) return rc;
  rc = plane2_buffers(c, NP, 0, res_bytes);
  if (rc) return rc;
  hipStream_t s = c->stream;
  const double t_first = host_now_ms();
  const size_t tstride = (size_t)(c->n_max + 1) * ld;
  {
    // results, per-plane corrections, used-feature mask (rounded up to whole words: the buffer is f_max + 64 bytes), flags,
    // [0] current T buffer + [1..2] factor bookkeeping (PlaneSolve::cond), half 0 of T (sum of the accepted L0^T A L0): one launch
    // both halves of T: a plane writes its candidate only inside its leading block, the rest of either half must read as zero;
    // the packed factor / inverted diagonal blocks behind the loop start out as the identity for the same reason
    const int ntn = (n + 15) / 16;
    void* zp[8] = {c->pl_res, c->pl_dx, c->pl_used, c->flags, c->pl_cur, c->pl_Tbuf, c->Ltp, c->Dinv};
    const size_t zb[8] = {sizeof(double) * 4 * NP, sizeof(double) * (size_t)n * NP, ((size_t)F + 3) & ~(size_t)3, sizeof(int) * 4,
                          3 * sizeof(int), sizeof(double) * (tstride + (size_t)n * ld),
                          c->pl_sub_active ? sizeof(double) * 256 * (size_t)(ntn * (ntn + 1) / 2) : 0,
                          c->pl_sub_active ? sizeof(double) * 256 * (size_t)ntn : 0};
    const int zpat[8] = {0, 0, 0, 0, 0, 0, 2, 1};
    HIPCHK(ovp_launch_fill_regions(zp, zb, zpat, 8, ntn, s));
  }
  if (c->pl_ktimer) {  // [0 | 1] = the whole loop on the device clock (first launch .. covariance product), then a pair per plane
    while (c->pl_ev_loop.size() < 2) {
      hipEvent_t e;
      HIPCHK(hipEventCreate(&e));
      c->pl_ev_loop.push_back(e);
    }
    if (!c->pl_sub_active) HIPCHK(hipEventRecord(c->pl_ev_loop[0], s));  // (plane_update_ordered: in front of its permutation)
  }
  // (a cheap look at the batch first: when no plane can qualify - update/UpdaterMSCKF.cpp:316-317, 384-396 - nothing below needs the
  // factor, and a singular prior must not fail a call that has nothing to update)
  bool any_candidate = false;
  {
    std::vector<int> cnt((size_t)NP + 1, 0);
    for (int f = 0; f < F; ++f) {
      const int pf = pb->plane_of_feat[f];
      if (pf >= 1 && pf <= NP && c->h_n_meas[f] >= 2) ++cnt[pf];
    }
    for (int q = 0; q < n_slam; ++q) ++cnt[pb->slam_plane[q]];
    for (int pl = 1; pl <= NP && !any_candidate; ++pl)
      any_candidate = cnt[pl] >= (pb->plane_state_id[pl - 1] >= 0 ? 1 : 4);
  }
  // L0 = chol(P), dense lower triangular in c->L.  Nothing needs it before the first plane's W = A L0, so it runs on the side
  // stream beside that plane's rows / Gram pair / assembly (round 5; one workgroup - the small kernels of the front end leave it a
  // CU on every XCD) and is joined in front of that product.  OVP_PL_CHOL_SIDE=0: on the loop's own stream, in front of everything.
  // (every way out of this function behind the fork - HIPCHK returns included - makes the loop's stream wait for the side stream:
  // the next call must not race a factorization that is still writing c->L / c->flags)
  struct ForkGuard {
    hipStream_t s;
    hipEvent_t ev_join;
    hipStream_t side;
    bool forked;
    ~ForkGuard() {
      if (!forked) return;
      (void)hipEventRecord(ev_join, side);  // (a second record behind whatever the side stream got: harmless when the first one made it)
      (void)hipStreamWaitEvent(s, ev_join, 0);
    }
  } fork_guard{s, c->ev_join, c->stream2, false};
  bool& chol_forked = fork_guard.forked;
  if (any_candidate) {
    const char* side_env = getenv("OVP_PL_CHOL_SIDE");  // (read per call: the tests switch it)
    if (!(side_env && side_env[0] == '0') && s == c->stream) {
      HIPCHK(hipEventRecord(c->ev_fork, s));
      HIPCHK(hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
      chol_forked = true;
      rc = chol_of_P(c, c->stream2);
      if (rc) return rc;
      HIPCHK(hipEventRecord(c->ev_join, c->stream2));
    } else {
      rc = chol_of_P(c, s);
      if (rc) return rc;
    }
  }
  auto join_chol = [&]() -> hipError_t {
    if (!chol_forked) return hipSuccess;
    chol_forked = false;
    return hipStreamWaitEvent(s, c->ev_join, 0);
  };
  // a refusal from here on: chol(P) has run - a flag it may have raised (singular prior) must not outlive the call
  auto bail = [&](int code) {
    (void)join_chol();
    (void)hipMemsetAsync(c->flags, 0, sizeof(int) * 4, s);
    return code;
  };
  // ---- host-side grouping (update/UpdaterMSCKF.cpp:204-229) ----
  std::vector<PlaneJobH> jobs;
  std::vector<int> featlist;
  std::vector<int> perms;  // per job: n entries
  const int ncal = (o->do_calib_camera_pose ? 6 : 0) + (o->do_calib_camera_intrinsics ? 8 : 0);
  // features bucketed by plane in one pass, batch order kept (a scan of the whole batch per plane was 0.4 ms of host time in
  // front of the first launch at 8000 features x 50 planes)
  std::vector<int> bucket_start((size_t)NP + 2, 0), bucket((size_t)(F > 0 ? F : 1));
  for (int f = 0; f < F; ++f) {
    const int pf = pb->plane_of_feat[f];
    if (pf >= 1 && pf <= NP) ++bucket_start[pf + 1];
  }
  for (int pl = 1; pl <= NP + 1; ++pl) bucket_start[pl] += bucket_start[pl - 1];
  {
    std::vector<int> fill(bucket_start.begin(), bucket_start.end());
    for (int f = 0; f < F; ++f) {
      const int pf = pb->plane_of_feat[f];
      if (pf >= 1 && pf <= NP) bucket[fill[pf]++] = f;
    }
  }
  for (int pl = 0; pl < NP; ++pl) {
    PlaneJobH j;
    j.pl = pl;
    j.start = (int)featlist.size();
    j.nf = 0;
    j.rows_total = 0;
    j.rows_live = 0;
    j.sid = pb->plane_state_id[pl];
    j.in_state = j.sid >= 0;
    unsigned long long seen = 0ull;
    for (int bi = bucket_start[pl + 1]; bi < bucket_start[pl + 2]; ++bi) {
      const int f = bucket[bi];
      const int m = c->h_n_meas[f];
      if (m < 2) continue;
      if (m > OVP_MAX_MEAS_DEV) return bail(OVP_E_CAPACITY);  // 2m bearing rows = one wavefront (the constraint row is wave-uniform)
      featlist.push_back(f);
      j.nf++;
      j.rows_total += 3 * m - 3;
      j.rows_live += 2 * m - 2;  // the m identical constraint rows are one direction (k_chol2 gate)
      for (int k = 0; k < m; ++k) seen |= 1ull << c->h_clone_idx[(size_t)f * M + k];
    }
    int ns_pl = 0;  // SLAM landmarks on this (out-of-state) plane: one row and three involved columns each
    if (!j.in_state)
      for (int q = 0; q < n_slam; ++q)
        if (pb->slam_plane[q] == pl + 1) ++ns_pl;
    if (ns_pl > PA_MAXQ) return bail(OVP_E_CAPACITY);
    j.ns_pl = ns_pl;
    if (j.nf == 0 || (!j.in_state && j.nf + ns_pl < 4)) {  // update/UpdaterMSCKF.cpp:316-317,384-396
      featlist.resize(j.start);
      continue;
    }
    j.rows_total += ns_pl;
    j.rows_live += ns_pl;
    const int c_ref = 6 * __builtin_popcountll(seen) + ncal + 3 * ns_pl;
    const int rows_c = j.rows_total > c_ref ? c_ref : j.rows_total;  // UpdaterPlane::measurement_compress_inplace
    j.rows_u = j.in_state ? rows_c : rows_c - 3;
    j.n_involved = c_ref + (j.in_state ? 3 : 0);
    if (!j.in_state) j.rows_total -= 3;
    if (!j.in_state) j.rows_live -= 3;
    if (j.rows_u < 1) {
      featlist.resize(j.start);
      continue;
    }
    j.thr = o->chi2_multiplier * ovp_chi2_quantile_095(j.rows_u);
    // order of the involved columns in the normalised Gram: everything that is not a clone first, the clones last (a rank
    // deficiency - gauge freedom, planar scene - then shows up in the trailing pivots, k_chol2 mode 2)
    {
      std::vector<int> perm(n, -1);
      int pos = 0;
      std::vector<char> inv(n, 0);
      if (o->do_calib_camera_pose)
        for (int k = 0; k < 6; ++k) inv[c->calib_id + k] = 1;
      if (o->do_calib_camera_intrinsics)
        for (int k = 0; k < 8; ++k) inv[c->intr_id + k] = 1;
      if (j.in_state)
        for (int k = 0; k < 3; ++k) inv[j.sid + k] = 1;
      if (!j.in_state)
        for (int q = 0; q < n_slam; ++q)
          if (pb->slam_plane[q] == pl + 1)
            for (int k = 0; k < 3; ++k) inv[pb->slam_state_id[q] + k] = 1;
      for (int col = 0; col < n; ++col)
        if (inv[col]) perm[col] = pos++;
      for (int ci = 0; ci < (int)c->h_clone_id.size(); ++ci)
        if ((seen >> ci) & 1ull)
          for (int k = 0; k < 6; ++k) {
            const int col = c->h_clone_id[ci] + k;
            if (col >= 0 && col < n && perm[col] < 0) perm[col] = pos++;
          }
      j.n_inv_cols = pos;
      perms.insert(perms.end(), perm.begin(), perm.end());
    }
    jobs.push_back(j);
  }
  const int NJ = (int)jobs.size();
  if (NJ == 0 && any_candidate) {  // chol(P)'s verdict concerns nobody
    HIPCHK(join_chol());
    HIPCHK(hipMemsetAsync(c->flags, 0, sizeof(int) * 4, s));
  }
  // ---- staging layout: ints [featlist | sid NP | perms NJ*n | slam_plane | slam_id], doubles [cp | cp_fej | slam_p | slam_p_fej] ----
  const size_t n_int = featlist.size() + (size_t)NP + perms.size() + 2 * (size_t)n_slam;
  const size_t int_bytes = ((n_int * sizeof(int) + 15) / 16) * 16;
  const size_t n_dbl = 6 * (size_t)NP + 6 * (size_t)n_slam;
  const size_t stage_bytes = int_bytes + n_dbl * sizeof(double);
  rc = plane2_buffers(c, NP, stage_bytes, res_bytes);  // (grows the staging block when this frame needs more)
  if (rc) return rc;
  int* hi = (int*)c->pl_hstage;
  double* hd = (double*)((char*)c->pl_hstage + int_bytes);
  int* di = (int*)c->pl_dstage;
  double* dd = (double*)((char*)c->pl_dstage + int_bytes);
  size_t io = 0;
  const size_t o_feat = io;
  memcpy(hi + io, featlist.data(), sizeof(int) * featlist.size());
  io += featlist.size();
  const size_t o_sid = io;
  memcpy(hi + io, pb->plane_state_id, sizeof(int) * NP);
  io += NP;
  const size_t o_perm = io;
  if (!perms.empty()) memcpy(hi + io, perms.data(), sizeof(int) * perms.size());
  io += perms.size();
  const size_t o_spl = io;
  if (n_slam) memcpy(hi + io, pb->slam_plane, sizeof(int) * n_slam);
  io += n_slam;
  const size_t o_sidx = io;
  if (n_slam) memcpy(hi + io, pb->slam_state_id, sizeof(int) * n_slam);
  io += n_slam;
  memcpy(hd, pb->cp, sizeof(double) * 3 * NP);
  memcpy(hd + 3 * NP, pb->cp_fej, sizeof(double) * 3 * NP);
  if (n_slam) {
    memcpy(hd + 6 * NP, pb->slam_p, sizeof(double) * 3 * n_slam);
    memcpy(hd + 6 * NP + 3 * n_slam, pb->slam_p_fej, sizeof(double) * 3 * n_slam);
  }
  HIPCHK(hipMemcpyAsync(c->pl_dstage, c->pl_hstage, stage_bytes, hipMemcpyHostToDevice, s));
  const int* d_feat = di + o_feat;
  const int* d_sid = di + o_sid;
  const int* d_perm = di + o_perm;
  const int* d_spl = di + o_spl;
  const int* d_sidx = di + o_sidx;
  double* d_cp = dd;
  double* d_cpfej = dd + 3 * NP;
  double* d_slam_p = dd + 6 * NP;
  double* d_slam_pfej = dd + 6 * NP + 3 * n_slam;
  const double white_c = 1.0 / o->sigma_constraint;
  // weight of the expected energy of the rounding-decided rows in the gate statistic (k_chol2.hip); OVP_PL_NOISE_SCALE overrides the
  // calibrated constant for the study that produced it (tools/plane_gate_agreement.py --fit)
  double noise_scale = OVP_PLANE_NOISE_KAPPA;
  if (const char* ns_env = getenv("OVP_PL_NOISE_SCALE")) noise_scale = atof(ns_env);  // (read per call)
  for (int jn = 0; jn < NJ; ++jn) {
    const PlaneJobH& j = jobs[jn];
    // leading block this plane's products and factorization run on (plane_update_ordered): every column involved so far
    const int nk = c->pl_sub_active ? c->pl_nl[j.pl] : n;
    // (1) per-feature rows
    ovp::PlaneParams pp;
    pp.feat_list = d_feat + j.start;
    pp.n_local = j.nf;
    pp.plane = j.pl;
    pp.in_state = j.in_state;
    pp.plane_sid = j.sid;
    pp.white_c = white_c;
    pp.cp = d_cp;
    pp.cp_fej = d_cpfej;
    pp.cst = c->pl_cst;
    ovp::FeatParams fpl = fp;
    fpl.n = nk;
    fpl.P = c->P;
    HIPCHK(ovp_launch_plane_feat(&fpl, &pp, j.nf, s));
    // (2) Gram products
    const int chunks = (2 * j.nf + c->rows_per_chunk - 1) / c->rows_per_chunk;
    int nsplit = 1;
    HIPCHK(ovp_launch_gram_pair(c->rec, fp.n_clones, j.nf, c->rows_per_chunk, chunks, c->gramS, c->G, 3 * j.nf, c->ldg, nk + 4,
                                c->n_split, c->part, &nsplit, s));
    // (3) pair on the state columns, normalised Gram, residual energy
    ovp::PlaneAsm pa;
    memset(&pa, 0, sizeof(pa));
    pa.gramS = c->gramS;
    pa.n_clones = fp.n_clones;
    pa.n_chunks = chunks;
    pa.part = c->part;
    pa.n_split = nsplit;
    {
      const int nt16 = (nk + 4 + 15) / 16;
      pa.ntile = nt16 * (nt16 + 1) / 2;
    }
    pa.colmap = c->colmap;
    pa.n = nk;
    pa.plane_sid = j.sid;
    pa.in_state = j.in_state;
    pa.cst = c->pl_cst;
    pa.nf = j.nf;
    pa.n_slam = j.in_state ? 0 : n_slam;
    pa.plane1 = j.pl + 1;
    pa.slam_plane = d_spl;
    pa.slam_id = d_sidx;
    pa.slam_p = d_slam_p;
    pa.slam_p_fej = d_slam_pfej;
    pa.cp = d_cp + 3 * j.pl;
    pa.cp_fej = d_cpfej + 3 * j.pl;
    pa.white_c = white_c;
    pa.do_fej = fp.do_fej;
    pa.Ab = c->Ab;
    pa.lda = ld;
    pa.perm = d_perm + (size_t)jn * n;
    pa.An = c->pl_An;
    pa.ldn = ld;
    pa.bn = c->pl_bn;
    pa.eps = 1e-12;
    pa.scal = c->pl_scal;
    HIPCHK(ovp_launch_plane_assemble2(&pa, s));
    // (4) W = A L0 ;  T_try = T_cur + L0^T W ;  c = L0^T b
    HIPCHK(join_chol());  // (first plane: L0 comes from the side stream)
    HIPCHK(ovp_launch_gemm4(0, 0, nk, nk, nk, c->Ab, ld, c->L, ld, c->W1, ld, 0, 0, s));
    HIPCHK(ovp_launch_plane_dT(nk, c->L, ld, c->W1, c->Ab + (size_t)nk * ld, c->pl_Tbuf, tstride, c->pl_cur, c->pl_crow, s));
    // (5) both factorizations, gate, solve, commit
    ovp::Chol2Job j0, j1;
    memset(&j0, 0, sizeof(j0));
    memset(&j1, 0, sizeof(j1));
    j0.A = c->pl_Tbuf;
    j0.sel = c->pl_cur;
    j0.sel_xor = 1;
    j0.sel_stride = tstride;
    j0.n = nk;
    j0.ld = ld;
    j0.add_identity = 1;
    j0.mode = 1;
    j0.brow = c->pl_crow;
    j0.flag = c->flags;
    j1.A = c->pl_An;
    j1.n = j.n_inv_cols;
    j1.ld = ld;
    j1.add_identity = 0;
    j1.mode = 2;
    j1.brow = c->pl_bn;
    j1.flag = c->flags + 2;
    j1.piv_floor = 1e-5;
    ovp::PlaneSolve ps;
    memset(&ps, 0, sizeof(ps));
    ps.scal = c->pl_scal;
    ps.range_done = c->pl_range_done;
    ps.seq = ++c->pl_seq;
    {
      // The update part on two workgroups: tile columns < h and the rest (k_chol2.hip).  Measured (r03, A/B in one call): at 16 tile
      // columns (N = 240) nothing is gained (2.91 against 2.81 ms per config-3 plane loop for h = 5 .. 8: exports + a second gate
      // hand-over cost what the second CU's f64 pipe gives), so one workgroup stays the default there; from 17 tile columns on
      // (N > 255) the tile registers of one workgroup spill and the split wins (config 4, N = 285: 7.91 ms for h = 5 or 6, 8.10 for
      // 7 or 8, 8.69 unsplit).  OVP_C2_SPLIT: 0 = never, h = forced.
      const char* split_s = getenv("OVP_C2_SPLIT");  // (read per call: the tests switch it)
      const int split_env = split_s ? atoi(split_s) : -1;
      const int nb = nk + 1, ntb = (nb + 15) / 16;
      const int nst = (nb % 16 == 1) ? ntb - 1 : ntb;  // a border row alone in its tile row takes no step
      int h = ntb >= 17 ? nst / 3 : 0;  // part B also runs the back half of the chain: 5 - 6 of 18 steps measured best (7.91 ms per
                                        // config-4 plane loop against 8.10 for 7 or 8 and 8.69 unsplit)
      if (split_env >= 0) h = split_env < ntb - 1 ? split_env : 0;
      if (h > 9) h = 9;  // pl_xbuf holds nine exported steps
      j0.split_h = h;
      j0.xbuf = c->pl_xbuf;
      j0.xflag = c->pl_xflag;
      j0.xseq = ps.seq;
      ps.xzz = c->pl_xy;
      ps.xy = c->pl_xy + 16;
      ps.xsync = c->pl_xflag + 32;
    }
    ps.thr = j.thr;
    ps.rows_live = j.rows_live;
    ps.rows_u = j.rows_u;
    ps.n_involved = j.n_inv_cols;
    ps.force = pb->force_decision ? (int)pb->force_decision[j.pl] : -1;
    ps.noise_scale = noise_scale;
    ps.tol_strict = 1e-5;
    ps.tol_loose = 1e-5;
    ps.res_out = c->pl_res + 4 * j.pl;
    ps.L0 = c->L;
    ps.ld0 = ld;
    ps.n_full = n;
    ps.dx_out = c->pl_dx + (size_t)j.pl * n;
    ps.dx_last = c->pl_dxlast;
    ps.cur = c->pl_cur;
    // The covariance product behind the loop needs the factor of the last ACCEPTED T.  The last few planes leave theirs behind when
    // they are accepted (~8 us of stores each); if one of them stays the last accepted plane, the k_tilechol behind the loop
    // (94 us at N = 240) finds nothing to do.  Which plane that is, is decided on the device.
    static const int emit_last = getenv("OVP_PL_EMIT_LAST") ? atoi(getenv("OVP_PL_EMIT_LAST")) : 4;
    ps.cond = c->pl_cur + 1;
    ps.seq_plane = jn + 1;
    ps.emit = (jn >= NJ - emit_last) ? 1 : 0;
    ps.Lpack = c->Ltp;
    ps.Dinv = c->Dinv;
    ps.feat_list = d_feat + j.start;
    ps.n_feat_local = j.nf;
    ps.feat_used = c->pl_used;
    ps.clone_R = c->clone_R;
    ps.clone_p = c->clone_p;
    ps.clone_id = c->clone_id;
    ps.n_clones = fp.n_clones;
    ps.cal = c->cal;
    ps.calib_id = o->do_calib_camera_pose ? c->calib_id : -1;
    ps.intr_id = o->do_calib_camera_intrinsics ? c->intr_id : -1;
    ps.cp = d_cp;
    ps.plane_sid = d_sid;
    ps.n_planes = NP;
    ps.n_slam = n_slam;
    ps.slam_id = d_sidx;
    ps.slam_p = d_slam_p;
    if (c->pl_ktimer == 1) {
      while ((int)c->pl_ev.size() < 2 * (jn + 1)) {
        hipEvent_t e;
        HIPCHK(hipEventCreate(&e));
        c->pl_ev.push_back(e);
      }
      HIPCHK(hipEventRecord(c->pl_ev[2 * jn], s));
    }
    static const bool pl_stamps = getenv("OVP_PL_STAMPS") != nullptr;  // diagnostics: cycle stamps of the last plane's tail
    static long long* d_stamps = nullptr;
    if (pl_stamps) {
      if (!d_stamps) HIPCHK(hipMalloc((void**)&d_stamps, sizeof(long long) * 2 * 16 * 32));
      j0.stamps = d_stamps;
    }
    HIPCHK(ovp_launch_chol2(&j0, &j1, &ps, s));
    if (c->pl_ktimer == 1) HIPCHK(hipEventRecord(c->pl_ev[2 * jn + 1], s));
    if (c->pl_sub_rest)
      HIPCHK(ovp_launch_plane_sub_accum(c->pl_res + 4 * j.pl, c->Ab, c->pl_Asum, c->pl_dx + (size_t)j.pl * n,
                                        c->pl_U + (size_t)j.pl * ld, nk, ld, s));
    if (pl_stamps && jn == NJ - 1) {
      long long h[2 * 16 * 32];
      HIPCHK(hipStreamSynchronize(s));
      HIPCHK(hipMemcpy(h, d_stamps, sizeof(h), hipMemcpyDeviceToHost));
      const int ntb = (nk + 1 + 15) / 16;
      const long long* e = h + (ntb + 1) * 16;
      fprintf(stderr, "[plane tail, cycles] factor %lld | gate %lld | back substitution %lld | dx = L0 y %lld | commit %lld\n",
              e[0] - h[0], e[1] - e[0], e[2] - e[1], e[3] - e[2], e[4] - e[3]);
      if (atoi(getenv("OVP_PL_STAMPS")) >= 2) {
        // per step, both parts of a split factorization, relative to part A's first stamp: elimination wave 0 [start | column
        // there | eliminated | signalled], tile wave 0 [start | panel there | next column updated | published | step done]
        const long long t0 = h[0];
        for (int part = 0; part < (j0.split_h > 0 ? 2 : 1); ++part) {
          const long long* hp = h + part * 16 * 32;
          fprintf(stderr, " part %c: prologue stamps %lld %lld %lld\n", part ? 'B' : 'A', hp[13] - t0, hp[14] - t0, hp[15] - t0);
          for (int k = 0; k < ntb; ++k) {
            const long long* q = hp + k * 16;
            if (!q[0] && !q[8]) continue;
            fprintf(stderr, "  k=%2d E %7lld %7lld %7lld %7lld | T %7lld %7lld %7lld %7lld %7lld\n", k, q[0] - t0, q[1] - t0, q[2] - t0,
                    q[3] - t0, q[8] - t0, q[9] - t0, q[10] - t0, q[11] - t0, q[12] - t0);
          }
          const long long* m = hp + (ntb + 1) * 16;
          fprintf(stderr, "  tail: factor done %lld, gate %lld, backsolve %lld, dx %lld, commit %lld\n", m[0] - t0, m[1] - t0, m[2] - t0,
                  m[3] - t0, m[4] - t0);
        }
      }
      HIPCHK(hipMemset(d_stamps, 0, sizeof(long long) * 2 * 16 * 32));
    }
  }
  // ---- the covariance, once:  P = L0 T^-1 L0^T = V^T V,  V = Lt^-1 L0^T ----
  bool factor_enqueued = false;
  if (NJ > 0) {
    {
      // chol of the accepted T (+ I) unless the last accepted plane left its factor behind; the second-generation kernel reads the
      // current half of the double buffer itself (Chol2Job::sel) - no copy into c->T in front of it
      const bool first_gen_T = getenv("OVP_TILECHOL_T") != nullptr;
      if (!first_gen_T && n <= ovp_chol2_max_n() + 1) {
        ovp::Chol2Job jt;
        memset(&jt, 0, sizeof(jt));
        jt.A = c->pl_Tbuf;
        jt.sel = c->pl_cur;
        jt.sel_xor = 0;
        jt.sel_stride = tstride;
        jt.n = n;
        jt.ld = ld;
        jt.add_identity = 1;
        jt.mode = 0;
        jt.flag = c->flags;
        jt.Lpack = c->Ltp;
        jt.Dinv_out = c->Dinv;
        jt.skip_cond = c->pl_cur + 1;
        HIPCHK(ovp_launch_chol2(&jt, nullptr, nullptr, s));
      } else {
        HIPCHK(ovp_launch_select_copy(c->T, c->pl_Tbuf, tstride, c->pl_cur, n, ld, 1, s));
        HIPCHK(chol_of_T(c, c->T, n, ld, 1, c->pl_cur + 1, s));
      }
    }
    HIPCHK(ovp_launch_fwdsub(c->Ltp, c->Dinv, c->L, c->Y, n, ld, 0, s));
    HIPCHK(ovp_launch_gemm4c(1, 0, n, n, n, c->Y, ld, c->Y, ld, c->P, ld, 0, 1, c->flags, s));
    // back into the state's own column order (unless a factorization failed: the resident P stays), and the factor of the
    // covariance just formed for the point update behind the loop (P = V^T V: M = V^T, rows in state order) - one launch for both
    const bool keep_factor = getenv("OVP_NO_KEPT_FACTOR") == nullptr;  // (read per call: the tests switch it)
    const bool want_factor = keep_factor && !c->pl_sub_rest && n <= OVP_TILECHOL_NMAX;
    if (want_factor && !c->Lkeep) HIPCHK(dalloc(&c->Lkeep, (size_t)c->n_max * ld));
    if (c->pl_scatter_dst) {
      HIPCHK(ovp_launch_unpermute_pair(c->P, c->Y, ld, c->pl_scatter_ids, n, c->pl_scatter_dst, want_factor ? c->Lkeep : nullptr, ld,
                                       c->flags, c->pl_boost_active ? c->boost_vec : nullptr, s));
      c->kept_boost = want_factor && c->pl_boost_active;  // Lkeep is a factor of P + diag(boost_vec): the point update on it
                                                          // takes the amounts off at its end (ekf_from_gram)
    } else if (want_factor) {
      HIPCHK(ovp_launch_factor_from_V(c->Y, ld, nullptr, n, c->Lkeep, ld, s));
    }
    factor_enqueued = want_factor;
  }
  if (c->pl_ktimer) HIPCHK(hipEventRecord(c->pl_ev_loop[1], s));
  // ---- results: one pinned block, one synchronisation ----
  double* hres = (double*)c->pl_hres;
  double* hdx = hres + 4 * (size_t)NP;
  unsigned char* hused = (unsigned char*)(hdx + (size_t)n * NP);
  HIPCHK(hipMemcpyAsync(hres, c->pl_res, sizeof(double) * 4 * NP, hipMemcpyDeviceToHost, s));
  if (dx_planes) HIPCHK(hipMemcpyAsync(hdx, c->pl_dx, sizeof(double) * (size_t)n * NP, hipMemcpyDeviceToHost, s));
  if (F) HIPCHK(hipMemcpyAsync(hused, c->pl_used, (size_t)F, hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(c->h_flags, c->flags, sizeof(int) * 4, hipMemcpyDeviceToHost, s));
  const double t_enq = host_now_ms();
  HIPCHK(hipStreamSynchronize(s));
  {
    c->host_acc[0] += t_first - t_entry;
    c->host_acc[1] += t_enq - t_entry;
    c->host_acc[2] += host_now_ms() - t_enq;
    c->host_acc[3] += 1.0;
  }
  if (c->pl_ktimer) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, c->pl_ev_loop[0], c->pl_ev_loop[1]) == hipSuccess) c->host_acc[7] += ms;
  }
  if (c->pl_ktimer == 1)
    for (int jn = 0; jn < NJ; ++jn) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, c->pl_ev[2 * jn], c->pl_ev[2 * jn + 1]) == hipSuccess) {
        c->pl_ktime_ms += ms;
        c->pl_klaunches += 1;
      }
    }
  c->pl_used_valid = true;
  c->h_pl_used.assign(hused, hused + F);
  if (dx_planes) memcpy(dx_planes, hdx, sizeof(double) * (size_t)n * NP);
  if (feat_used && F) memcpy(feat_used, hused, (size_t)F);
  for (const PlaneJobH& j : jobs) {
    if (plane_ok) plane_ok[j.pl] = hres[4 * j.pl + 1] > 0.5 ? 1 : 0;
    if (plane_chi2) plane_chi2[j.pl] = hres[4 * j.pl];
    if (plane_dof) plane_dof[j.pl] = j.rows_u;
  }
  const int bad = c->h_flags[0] | c->h_flags[2];
  HIPCHK(hipMemsetAsync(c->flags, 0, sizeof(int) * 4, s));
  if (bad) {
    // A failed factorization / timed-out hand-over rejects its plane and every later one before anything is committed, and the
    // covariance product behind the loop is cancelled (the resident P is the prior).  Planes accepted BEFORE the failure have
    // committed their corrections to the device tables: those no longer belong to the resident covariance - the caller must
    // upload the state again (OVP_E_STATE until then).  chol(P) itself failing (singular prior) happens in front of every plane.
    bool any_committed = false;
    for (const PlaneJobH& j : jobs) any_committed |= hres[4 * j.pl + 1] > 0.5;
    if (any_committed) c->have_state = false;
    if (bad == 1 && !any_committed && !c->pl_psd) {
      // chol(P) hit a non-positive pivot: the prior is only positive SEMI-definite.  Nothing was committed and the resident
      // covariance was not written - the same loop once more on the pivot-dropping factor (chol_of_P).
      c->pl_psd = true;
      const int rc2 = ovp_msckf_plane_update(c, o, pb, dx_planes, plane_ok, plane_chi2, plane_dof, feat_used);
      c->pl_psd = false;
      return rc2;
    }
  }
  if (bad & 2) return OVP_E_TIMEOUT;
  if (bad) return OVP_E_NOTSPD;
  c->have_factor = factor_enqueued;
  return 0;
}

// ---- UpdaterPlane::init_vio_plane core ----------------------------------------------------------
extern "C" int ovp_plane_init(ovp_ctx* c, const ovp_update_opts* o, const ovp_plane_batch* pb, double const_init_multi,
                              double const_init_chi2, double* dx_planes, int dx_stride, uint8_t* plane_ok, double* plane_chi2,
                              int* plane_dof, int* new_ids, double* cp_new, uint8_t* feat_used) {
  drop_kept_factor(c);  // (writes the covariance: a kept factor no longer belongs to it)
  if (!c || !o || !pb || pb->n_planes < 0) return OVP_E_ARG;
  if (!c->have_state || !c->have_cov || !c->have_batch) return OVP_E_STATE;
  if (c->h_n_meas.empty() && c->n_feats > 0) return OVP_E_STATE;
  const int ld = c->ld, F = c->n_feats, NP = pb->n_planes, M = c->max_meas;
  if (feat_used) memset(feat_used, 0, (size_t)F);
  for (int pl = 0; pl < NP; ++pl) {
    if (plane_ok) plane_ok[pl] = 0;
    if (plane_chi2) plane_chi2[pl] = 0.0;
    if (plane_dof) plane_dof[pl] = 0;
    if (new_ids) new_ids[pl] = -1;
    if (cp_new) memcpy(cp_new + 3 * pl, pb->cp + 3 * pl, 3 * sizeof(double));
    if (dx_planes) memset(dx_planes + (size_t)pl * dx_stride, 0, sizeof(double) * dx_stride);
  }
  if (NP == 0) return 0;
  c->pl_n_slam = 0;
  int rc = fill_feat_params(c, o);
  if (rc) return rc;
  rc = plane_buffers(c, NP);
  if (rc) return rc;
  rc = plane2_buffers(c, NP, 0, 0);  // (pl_crow: scale vector of the pivot-dropping factor, chol_of_P on a semi-definite prior)
  if (rc) return rc;
  hipStream_t s = c->stream;
  const int ncal = (o->do_calib_camera_pose ? 6 : 0) + (o->do_calib_camera_intrinsics ? 8 : 0);
  std::vector<int> sid(NP, -1);
  HIPCHK(hipMemcpyAsync(c->pl_sid, sid.data(), sizeof(int) * NP, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(c->pl_cp, pb->cp, sizeof(double) * 3 * NP, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(c->pl_cp_fej, pb->cp, sizeof(double) * 3 * NP, hipMemcpyHostToDevice, s));
  HIPCHK(hipStreamSynchronize(s));
  std::vector<double> res4(4), dxh(c->n_max), dcp(3);
  bool psd_prior = false;
  // The plane runs on the MARGINAL of the columns its rows can touch (clones and calibration: at most 6 * 32 + 14 of them) and the
  // rest of the state follows from the push-through identity, like the plane loop's sub-state (plane_update_ordered):
  // update/UpdaterPlane.cpp:296-481 has no size limit, the factorizations are those of ~80 columns instead of the state's, and the
  // marginal of a prior with an exact stochastic clone is positive definite (no second attempt).  Closed-loop session with two
  // planes, per frame: 0.198 ms against 0.216 on the whole state.  OVP_PLANE_INIT_SUB=0: the whole state (<= 288 columns; A/B, tests).
  const char* sub_env = getenv("OVP_PLANE_INIT_SUB");  // (read per call)
  const bool whole_state = sub_env && sub_env[0] == '0' && c->n <= OVP_TILECHOL_NMAX;
  std::vector<int> sub_ids, sub_pos;
  SubTables sub_t;
  int ns = 0;
  if (!whole_state) {
    const int n = c->n;
    sub_pos.assign((size_t)c->n_max, -1);
    bool bad_id = false;
    auto place = [&](int id, int sz) {
      if (id < 0 || id + sz > n) {
        bad_id = true;
        return;
      }
      for (int k = 0; k < sz; ++k)
        if (sub_pos[id + k] < 0) {
          sub_pos[id + k] = (int)sub_ids.size();
          sub_ids.push_back(id + k);
        }
    };
    for (int i = 0; i < c->fp.n_clones; ++i) place(c->h_clone_id[i], 6);
    if (o->do_calib_camera_pose) place(c->calib_id, 6);
    if (o->do_calib_camera_intrinsics) place(c->intr_id, 8);
    if (bad_id) return OVP_E_ARG;
    ns = (int)sub_ids.size();
    if (ns > OVP_TILECHOL_NMAX) return OVP_E_CAPACITY;
    rc = sub_tables_upload(c, o, sub_ids, sub_pos, &sub_t, s);
    if (rc) return rc;
    if (!c->pl_Asum) HIPCHK(dalloc(&c->pl_Asum, (size_t)c->n_max * ld));
    if (c->pl_U_cap < 1) {
      c->pl_U_cap = 8;
      HIPCHK(dalloc(&c->pl_U, (size_t)c->pl_U_cap * ld));
    }
  }
  for (int pl = 0; pl < NP; ++pl) {
    const int n = c->n;
    if ((n > OVP_TILECHOL_NMAX && !ns) || n + 3 > c->n_max) return OVP_E_CAPACITY;
    std::vector<int> featlist;
    int rows_total = 0, rows_live = 0;
    unsigned long long seen = 0ull;
    for (int f = 0; f < F; ++f) {
      if (pb->plane_of_feat[f] != pl + 1) continue;
      const int m = c->h_n_meas[f];
      if (m < 2) continue;
      if (m > OVP_MAX_MEAS_DEV) return OVP_E_CAPACITY;
      featlist.push_back(f);
      rows_total += 3 * m - 3;
      rows_live += 2 * m - 2;
      for (int k = 0; k < m; ++k) seen |= 1ull << c->h_clone_idx[(size_t)f * M + k];
    }
    const int nf = (int)featlist.size();
    if (nf < 3) continue;  // update/UpdaterPlane.cpp:303
    const int c_ref = 6 * __builtin_popcountll(seen) + ncal;
    const int rows_c = rows_total > c_ref ? c_ref : rows_total;
    if (rows_c - 3 < 1) continue;
    // the chi2 of StateHelper::initialize covers the rows that do not involve the plane, with dof = all rows (:471)
    const double thr = const_init_chi2 * ovp_chi2_quantile_095(rows_c);
    HIPCHK(hipMemcpyAsync(c->pl_featlist, featlist.data(), sizeof(int) * nf, hipMemcpyHostToDevice, s));
    ovp::FeatParams fp = c->fp;
    // A prior that is only positive SEMI-definite (an exact stochastic clone in front of the next propagation: every frame of a
    // running filter) fails chol(P) before anything is committed: the plane runs once more on the pivot-dropping factor of the
    // unit-diagonal form (chol_of_P, as the plane loop does), and so do the planes behind it.
    const int nj = ns ? ns : n;  // the size the plane's kernels run on
    SubSaved sub_sv;
    if (ns) {  // the marginal of the selection stands in for the state
      HIPCHK(ovp_launch_gather_block(c->P, ld, sub_t.d_ids, ns, c->P_tmp, ld, s));
      sub_sv = sub_enter(c, sub_t, ns, c->P_tmp);
      rc = fill_feat_params(c, o);  // (the calibration columns of the selection)
      if (rc) {
        sub_leave(c, sub_sv);
        (void)fill_feat_params(c, o);
        return rc;
      }
      fp = c->fp;
    }
    for (int attempt = 0; attempt < 2; ++attempt) {
      c->pl_psd = psd_prior;
      rc = (int)hipMemsetAsync(c->flags, 0, sizeof(int) * 4, s);
      if (!rc) rc = (int)hipMemsetAsync(c->pl_res + 4 * pl, 0, sizeof(double) * 4, s);
      if (!rc) rc = chol_of_P(c, s);
      if (!rc)
        rc = plane_job_device(c, o, fp, pl, 0, nf, 0, -1, 1.0 / (const_init_multi * o->sigma_constraint), c->L, 0, thr, rows_live - 3,
                              rows_c - 3, c_ref);
      c->pl_psd = false;
      if (!rc) rc = (int)hipMemcpyAsync(res4.data(), c->pl_res + 4 * pl, sizeof(double) * 4, hipMemcpyDeviceToHost, s);
      if (!rc) rc = (int)hipMemcpyAsync(c->h_flags, c->flags, sizeof(int) * 4, hipMemcpyDeviceToHost, s);
      if (!rc) rc = (int)hipStreamSynchronize(s);
      if (rc) break;  // (the selection's tables are taken off below before the error goes out)
      if (!c->h_flags[0] || psd_prior || nj > ovp_chol2_max_n() + 1) break;
      psd_prior = true;
    }
    if (ns) {
      sub_leave(c, sub_sv);
      const int rf = fill_feat_params(c, o);
      if (!rc) rc = rf;
      fp = c->fp;
    }
    if (rc) return rc;
    if (c->h_flags[0]) {
      (void)hipMemsetAsync(c->flags, 0, sizeof(int) * 4, s);
      return OVP_E_NOTSPD;
    }
    if (plane_chi2) plane_chi2[pl] = res4[0];
    if (plane_dof) plane_dof[pl] = rows_c;
    if (res4[1] < 0.5) continue;  // chi2 rejected: StateHelper::initialize returns false
    // accepted: P <- P+ = V^T V, append the plane, update the device tables like Type::update would
    if (ns) {
      // the selection's posterior Pss+ = V^T V; the whole state by the push-through identity (G = P[:, s], A|b = the plane's pair on s):
      //   u = b - A dx_s,  dx = G u ;  Lambda = A - A Pss+ A,  P -= G Lambda G^T
      HIPCHK(ovp_launch_gemm4(1, 0, ns, ns, ns, c->Y, ld, c->Y, ld, c->P_tmp, ld, 0, 1, s));
      HIPCHK(hipMemsetAsync(c->pl_Asum, 0, sizeof(double) * (size_t)ns * ld, s));
      HIPCHK(ovp_launch_plane_sub_accum(c->pl_res + 4 * pl, c->Ab, c->pl_Asum, c->dx, c->pl_U, ns, ld, s));  // (Asum = A from here)
      HIPCHK(ovp_launch_gemm4(0, 0, ns, ns, ns, c->pl_Asum, ld, c->P_tmp, ld, c->W1, ld, 0, 0, s));
      HIPCHK(ovp_launch_gemm4(0, 0, ns, ns, ns, c->W1, ld, c->pl_Asum, ld, c->T, ld, 0, 1, s));
      HIPCHK(ovp_launch_mat_sub(c->pl_Asum, c->T, c->T, ns, ns, ld, s));
      HIPCHK(ovp_launch_gather_cols(c->P, ld, sub_t.d_ids, n, ns, c->Y, ld, s));
      HIPCHK(ovp_launch_gemm4(0, 1, 1, n, ns, c->pl_U, ld, c->Y, ld, c->Lt, ld, 0, 0, s));
      HIPCHK(hipMemcpyAsync(c->dx, c->Lt, sizeof(double) * n, hipMemcpyDeviceToDevice, s));
      HIPCHK(ovp_launch_gemm4(0, 0, n, ns, ns, c->Y, ld, c->T, ld, c->W1, ld, 0, 0, s));
      HIPCHK(ovp_launch_gemm4(0, 1, n, n, ns, c->W1, ld, c->Y, ld, c->L, ld, 0, 1, s));
      HIPCHK(ovp_launch_sub_sym(c->P, c->L, n, ld, s));
      HIPCHK(ovp_launch_plane_init_augment(c->pl_E, c->ldg, ns, sub_t.d_ids, n, c->P, ld, c->dx, c->pl_scal + 4, s));
    } else {
      HIPCHK(ovp_launch_gemm4(1, 0, n, n, n, c->Y, ld, c->Y, ld, c->P, ld, 0, 1, s));
      HIPCHK(ovp_launch_plane_init_augment(c->pl_E, c->ldg, n, nullptr, n, c->P, ld, c->dx, c->pl_scal + 4, s));
    }
    HIPCHK(ovp_launch_plane_commit(c->pl_res + 4 * pl, nullptr /* no factor is chained here */, nullptr, n, ld, c->dx,
                                   c->pl_dx + (size_t)pl * c->n_max, c->clone_R, c->clone_p, c->clone_id, fp.n_clones, c->cal,
                                   o->do_calib_camera_pose ? c->calib_id : -1, o->do_calib_camera_intrinsics ? c->intr_id : -1,
                                   c->pl_cp, c->pl_sid, 0, s));
    HIPCHK(hipMemcpyAsync(dxh.data(), c->dx, sizeof(double) * n, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(dcp.data(), c->pl_scal + 4, sizeof(double) * 3, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    c->n = n + 3;
    if (plane_ok) plane_ok[pl] = 1;
    if (new_ids) new_ids[pl] = n;
    if (cp_new)
      for (int k = 0; k < 3; ++k) cp_new[3 * pl + k] = pb->cp[3 * pl + k] + dcp[k];
    if (dx_planes) memcpy(dx_planes + (size_t)pl * dx_stride, dxh.data(), sizeof(double) * (n < dx_stride ? n : dx_stride));
    if (feat_used)
      for (int f : featlist) feat_used[f] = 1;
  }
  return 0;
}

// ---- StateHelper::EKFUpdate with a dense host H ------------------------------------------------
extern "C" int ovp_ekf_update(ovp_ctx* c, const double* H_host, int rows, int cols, int ld, const int* col_ids,
                              const double* res_host, double* dx_host, ovp_update_info* info) {
  drop_kept_factor(c);  // (writes the covariance: a kept factor no longer belongs to it)
  if (!c || !H_host || !col_ids || !res_host || rows < 1 || cols < 1 || ld < rows) return OVP_E_ARG;
  if (!c->have_cov) return OVP_E_STATE;
  if (cols > c->n) return OVP_E_ARG;
  const int n = c->n;
  for (int j = 0; j < cols; ++j)
    if (col_ids[j] < 0 || col_ids[j] >= n) return OVP_E_ARG;
  // few rows (a frame's landmark re-observations, a zero-velocity update): the reference's own S-form on the kernels of
  // csrc/k_init.hip - S = H P H^T + I in LDS, P+ = P - W W^T - instead of two N x N factorizations
  const char* form_env = getenv("OVP_EKF_INFO_FORM");  // read per call: the tests run both forms in one process
  const bool info_form_only = form_env && form_env[0] == '1';
  if (!info_form_only && rows <= ovp_init_max_rows() && ovp_init_core_lds(0, rows, cols) <= ovp_init_max_lds() && cols <= c->n_max) {
    hipStream_t s = c->stream;
    const size_t oHt = 0, oRes = oHt + (size_t)cols * rows, oId = oRes + rows + 8;
    const size_t bytes = oId * sizeof(double) + sizeof(int) * (size_t)cols + 64;
    const size_t res_doubles = 4 + (size_t)c->n_max + 8;
    int rc = plane2_buffers(c, 0, bytes, res_doubles * sizeof(double));
    if (rc) return rc;
    double* h = (double*)c->pl_hstage;
    double* d = (double*)c->pl_dstage;
    for (int a = 0; a < cols; ++a) memcpy(h + oHt + (size_t)a * rows, H_host + (size_t)a * ld, sizeof(double) * rows);  // = H^T row-major
    memcpy(h + oRes, res_host, sizeof(double) * rows);
    memcpy(h + oId, col_ids, sizeof(int) * cols);
    const int* did = (const int*)(d + oId);
    double* dres = c->smallbuf;
    double* dM = dres + res_doubles;
    double* dLi = dM + (size_t)n * rows;
    double* dy = dLi + (size_t)rows * rows;
    if ((size_t)(dy + rows + 8 - c->smallbuf) > c->small_cap) return OVP_E_CAPACITY;
    HIPCHK(hipMemcpyAsync(c->pl_dstage, c->pl_hstage, bytes, hipMemcpyHostToDevice, s));
    HIPCHK(ovp_launch_init_m(c->P, c->ld, n, did, cols, d + oHt, rows, dM, s));
    HIPCHK(ovp_launch_init_core(c->P, c->ld, n, did, cols, d + oHt, 0, rows, dM, d + oRes /* unused: k = 0 */, d + oRes, d + oRes, 1.0,
                                1e300, dLi, dy, dres, s));
    HIPCHK(ovp_launch_init_update(c->P, c->P_tmp, c->ld, n, dM, rows, 0, rows, dLi, dy, dres, dres + 4, s));
    double* hres = (double*)c->pl_hres;
    HIPCHK(hipMemcpyAsync(hres, dres, sizeof(double) * (4 + (size_t)n), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (info) {
      memset(info, 0, sizeof(*info));
      info->n_rows = rows;
      info->n_cols = cols;
      info->not_spd = hres[1] > 0.5 ? 0 : 1;
      info->neg_diag = hres[2] != 0.0;
    }
    if (!(hres[1] > 0.5)) return OVP_E_NOTSPD;  // S = H P H^T + I lost definiteness: P is not a covariance; nothing was written
    double* t = c->P;
    c->P = c->P_tmp;
    c->P_tmp = t;
    if (dx_host) memcpy(dx_host, hres + 4, sizeof(double) * n);
    return hres[2] != 0.0 ? OVP_E_NEGDIAG : 0;
  }
  const size_t need = (size_t)ld * cols;
  if (need > c->Hd_cap) {
    if (c->Hd) hipFree(c->Hd);
    HIPCHK(dalloc(&c->Hd, need));
    c->Hd_cap = need;
  }
  if ((size_t)rows > c->res_cap) {
    if (c->resd) hipFree(c->resd);
    HIPCHK(dalloc(&c->resd, (size_t)rows + 64));
    c->res_cap = (size_t)rows + 64;
  }
  if (!c->Acc) HIPCHK(dalloc(&c->Acc, (size_t)c->n_max * c->n_max));
  if (!c->bcc) HIPCHK(dalloc(&c->bcc, (size_t)c->n_max));
  HIPCHK(hipMemsetAsync(c->flags, 0, sizeof(int) * 4, c->stream));
  HIPCHK(hipMemcpyAsync(c->Hd, H_host, sizeof(double) * need, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(c->resd, res_host, sizeof(double) * rows, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(c->idbuf, col_ids, sizeof(int) * cols, hipMemcpyHostToDevice, c->stream));
  // column-major H [rows x cols, ld] == row-major H^T [cols x rows, ld]:  A = H^T H, b = H^T r
  HIPCHK(ovp_launch_gemm(0, 1, cols, cols, rows, c->Hd, ld, c->Hd, ld, c->Acc, cols, 0, c->stream));
  HIPCHK(ovp_launch_gemm(0, 0, cols, 1, rows, c->Hd, ld, c->resd, 1, c->bcc, 1, 0, c->stream));
  HIPCHK(hipMemsetAsync(c->Ab, 0, sizeof(double) * (size_t)(n + 1) * c->ld, c->stream));
  HIPCHK(ovp_launch_scatter_gram(c->Acc, c->bcc, cols, c->idbuf, c->Ab, c->ld, n, c->stream));
  {
    std::vector<int> ids(col_ids, col_ids + cols);
    std::sort(ids.begin(), ids.end());
    ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
    int rs = set_substate(c, ids);
    if (rs) return rs;
  }
  int rc = ekf_from_gram(c, false);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(c->h_dx, c->dx, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync(c->h_flags, c->flags, sizeof(int) * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (c->h_flags[0]) {  // positive semi-definite prior: S-form instead of the factor of P
    int rs = ekf_sform(c);
    if (rs) return rs;
  }
  if (dx_host) memcpy(dx_host, c->h_dx, sizeof(double) * n);
  if (info) {
    memset(info, 0, sizeof(*info));
    info->n_rows = rows;
    info->n_cols = cols;
    info->not_spd = c->h_flags[0];
    info->neg_diag = c->h_flags[1];
  }
  if (c->h_flags[0]) return OVP_E_NOTSPD;
  if (c->h_flags[1]) return OVP_E_NEGDIAG;
  return 0;
}

// ---- UpdaterSLAM::update on the device (update/UpdaterSLAM.cpp:424-673; csrc/k_slam.hip) ------------------------------------
// Rows and gate of every landmark in ONE launch against the resident covariance (no download of P, no host gate), the accepted rows
// stacked on the device, StateHelper::EKFUpdate on that stack (S-form up to 80 rows, information form above), one synchronisation.
extern "C" int ovp_slam_update(ovp_ctx* c, const ovp_update_opts* o, const ovp_slam_batch* b, double* dx_host, uint8_t* status_host,
                               double* chi2_host, ovp_update_info* info) {
  drop_kept_factor(c);  // (writes the covariance: a kept factor no longer belongs to it)
  if (!c || !o || !b || b->n_landmarks < 0) return OVP_E_ARG;
  if (!c->have_state || !c->have_cov) return OVP_E_STATE;
  const int L = b->n_landmarks, n = c->n, M = b->max_meas;
  if (info) memset(info, 0, sizeof(*info));
  if (dx_host) memset(dx_host, 0, sizeof(double) * n);
  if (L == 0) return 0;
  if (M < 1 || M > OVP_MAX_MEAS || !b->n_meas || !b->landmark_id) return OVP_E_ARG;
  const bool any_pre = b->pre_rows != nullptr;
  if (any_pre && (!b->pre_cols || !b->pre_H || !b->pre_ids)) return OVP_E_ARG;
  const unsigned calmask = (o->do_calib_camera_pose ? 0x3Fu : 0u) | (o->do_calib_camera_intrinsics ? (0xFFu << 6) : 0u);
  int calcol[14];
  for (int k = 0; k < 14; ++k) {
    calcol[k] = (k < 6) ? c->calib_id + k : c->intr_id + (k - 6);
    if (!((calmask >> k) & 1)) calcol[k] = 0;
    else if (calcol[k] < 0 || calcol[k] >= n) return OVP_E_ARG;
  }
  const int C = (int)c->h_clone_id.size();
  // ---- host: the call's column list (first-seen order, as Hx_order_big of :634-646), row offsets, kernel geometry
  std::vector<int> gpos(n, -1), gids, row0(L), pre_off(L, 0), pre_ids_off(L, 0);
  auto touch = [&](int col) {
    if (gpos[col] < 0) {
      gpos[col] = (int)gids.size();
      gids.push_back(col);
    }
  };
  int m_total = 0, rows_max = 1, cols_max = 1;
  size_t preH = 0, preI = 0;
  bool any_built = false;
  for (int l = 0; l < L; ++l) {
    row0[l] = m_total;
    int rows, cols;
    if (any_pre && b->pre_rows[l] > 0) {
      rows = b->pre_rows[l];
      cols = b->pre_cols[l];
      if (cols < 1 || cols > n) return OVP_E_ARG;
      pre_off[l] = (int)preH;
      pre_ids_off[l] = (int)preI;
      for (int k = 0; k < cols; ++k) {
        const int id = b->pre_ids[preI + k];
        if (id < 0 || id >= n) return OVP_E_ARG;
        touch(id);
      }
      preH += (size_t)rows * cols + rows;
      preI += cols;
    } else {
      if (!b->uv || !b->clone_idx || !b->p_FinG || !b->p_FinG_fej) return OVP_E_ARG;
      const int m = b->n_meas[l];
      if (m < 0 || m > M) return OVP_E_ARG;
      const bool plane = b->plane_state_id && b->plane_state_id[l] >= 0;
      if (plane && (!b->cp || !b->cp_fej || b->plane_state_id[l] + 3 > n)) return OVP_E_ARG;
      if (b->landmark_id[l] < 0 || b->landmark_id[l] + 3 > n) return OVP_E_ARG;
      rows = plane ? 3 * m : 2 * m;
      cols = 6 * m + __builtin_popcount(calmask) + 3 + (plane ? 3 : 0);
      for (int a = 0; a < m; ++a) {
        const int ci = b->clone_idx[(size_t)l * M + a];
        if (ci < 0 || ci >= C) return OVP_E_ARG;
        for (int k = 0; k < 6; ++k) touch(c->h_clone_id[ci] + k);
      }
      if (m > 0) {
        for (int k = 0; k < 14; ++k)
          if ((calmask >> k) & 1) touch(calcol[k]);
        for (int k = 0; k < 3; ++k) touch(b->landmark_id[l] + k);
        if (plane)
          for (int k = 0; k < 3; ++k) touch(b->plane_state_id[l] + k);
      }
      any_built = any_built || m > 0;
    }
    m_total += rows;
    rows_max = std::max(rows_max, rows);
    cols_max = std::max(cols_max, cols);
  }
  if (m_total < 1) {  // nothing to update with (:661-663)
    if (status_host) memset(status_host, 0, L);
    if (chi2_host) memset(chi2_host, 0, sizeof(double) * L);
    return 0;
  }
  const int gcols = (int)gids.size();
  if (ovp_slam_gate_lds(rows_max, cols_max, 0) > 150 * 1024) return OVP_E_CAPACITY;
  const int h_in_lds = ovp_slam_gate_lds(rows_max, cols_max, 1) <= 150 * 1024 ? 1 : 0;
  hipStream_t s = c->stream;
  // ---- one pinned staging block -> one copy
  auto al = [](size_t v) { return (v + 63) & ~(size_t)63; };
  size_t off = 0;
  auto take = [&](size_t bytes) {
    const size_t o0 = off;
    off = al(off + bytes);
    return o0;
  };
  const size_t o_p = take(sizeof(double) * 3 * L), o_pf = take(sizeof(double) * 3 * L), o_cp = take(sizeof(double) * 3 * L),
               o_cpf = take(sizeof(double) * 3 * L), o_preH = take(sizeof(double) * (preH + 1)), o_uv = take(sizeof(float) * 2 * (size_t)L * M),
               o_ci = take(sizeof(int) * (size_t)L * M), o_nm = take(sizeof(int) * L), o_lm = take(sizeof(int) * L),
               o_ps = take(sizeof(int) * L), o_r0 = take(sizeof(int) * L), o_gp = take(sizeof(int) * n),
               o_gi = take(sizeof(int) * gcols), o_pr = take(sizeof(int) * L), o_pc = take(sizeof(int) * L),
               o_po = take(sizeof(int) * L), o_pio = take(sizeof(int) * L), o_pid = take(sizeof(int) * (preI + 1));
  const size_t stage_bytes = off;
  const size_t res_doubles = 4 + (size_t)c->n_max + 8;
  const size_t lres_bytes = al(sizeof(double) * L) + al((size_t)L);
  int rc = plane2_buffers(c, 0, stage_bytes, res_doubles * sizeof(double) + lres_bytes + 64);
  if (rc) return rc;
  char* h = (char*)c->pl_hstage;
  char* d = (char*)c->pl_dstage;
  memset(h, 0, stage_bytes);
  if (any_built) {
    memcpy(h + o_p, b->p_FinG, sizeof(double) * 3 * L);
    memcpy(h + o_pf, b->p_FinG_fej, sizeof(double) * 3 * L);
    memcpy(h + o_uv, b->uv, sizeof(float) * 2 * (size_t)L * M);
    memcpy(h + o_ci, b->clone_idx, sizeof(int) * (size_t)L * M);
  }
  if (b->cp) memcpy(h + o_cp, b->cp, sizeof(double) * 3 * L);
  if (b->cp_fej) memcpy(h + o_cpf, b->cp_fej, sizeof(double) * 3 * L);
  memcpy(h + o_nm, b->n_meas, sizeof(int) * L);
  memcpy(h + o_lm, b->landmark_id, sizeof(int) * L);
  for (int l = 0; l < L; ++l) ((int*)(h + o_ps))[l] = b->plane_state_id ? b->plane_state_id[l] : -1;
  memcpy(h + o_r0, row0.data(), sizeof(int) * L);
  memcpy(h + o_gp, gpos.data(), sizeof(int) * n);
  memcpy(h + o_gi, gids.data(), sizeof(int) * gcols);
  if (any_pre) {
    memcpy(h + o_pr, b->pre_rows, sizeof(int) * L);
    memcpy(h + o_pc, b->pre_cols, sizeof(int) * L);
    memcpy(h + o_po, pre_off.data(), sizeof(int) * L);
    memcpy(h + o_pio, pre_ids_off.data(), sizeof(int) * L);
    memcpy(h + o_preH, b->pre_H, sizeof(double) * preH);
    memcpy(h + o_pid, b->pre_ids, sizeof(int) * preI);
  }
  // ---- device buffers: the stacked system (Hd = H^T [gcols][m_total], resd), per-landmark results, block scratch
  const size_t need = (size_t)gcols * m_total;
  if (need > c->Hd_cap) {
    if (c->Hd) hipFree(c->Hd);
    c->Hd = nullptr;
    c->Hd_cap = 0;
    HIPCHK(dalloc(&c->Hd, need + 64));
    c->Hd_cap = need + 64;
  }
  if ((size_t)m_total > c->res_cap) {
    if (c->resd) hipFree(c->resd);
    c->resd = nullptr;
    c->res_cap = 0;
    HIPCHK(dalloc(&c->resd, (size_t)m_total + 64));
    c->res_cap = (size_t)m_total + 64;
  }
  if (lres_bytes > c->slam_res_cap) {
    if (c->slam_res) hipFree(c->slam_res);
    c->slam_res = nullptr;
    c->slam_res_cap = 0;
    HIPCHK(hipMalloc(&c->slam_res, lres_bytes + 4096));
    c->slam_res_cap = lres_bytes + 4096;
  }
  if (!h_in_lds) {
    const size_t hs = (size_t)L * rows_max * cols_max;
    if (hs > c->slam_hscr_cap) {
      if (c->slam_hscr) hipFree(c->slam_hscr);
      c->slam_hscr = nullptr;
      c->slam_hscr_cap = 0;
      HIPCHK(dalloc(&c->slam_hscr, hs + 64));
      c->slam_hscr_cap = hs + 64;
    }
  }
  HIPCHK(hipMemcpyAsync(d, h, stage_bytes, hipMemcpyHostToDevice, s));
  ovp::SlamParams sp;
  memset(&sp, 0, sizeof(sp));
  sp.fp = c->fp;
  sp.fp.uv = (const float*)(d + o_uv);
  sp.fp.clone_idx = (const int*)(d + o_ci);
  sp.fp.n_meas = (const int*)(d + o_nm);
  sp.fp.p_FinG = (const double*)(d + o_p);
  sp.fp.n_feats = L;
  sp.fp.max_meas = M;
  sp.fp.do_fej = o->do_fej;
  sp.fp.calmask = calmask;
  for (int k = 0; k < 14; ++k) sp.fp.calcol[k] = calcol[k];
  sp.fp.white_px = 1.0 / o->sigma_px;
  sp.fp.chi2_mult = o->chi2_multiplier;
  sp.fp.chi2_table = c->chi2_table;
  sp.fp.P = c->P;
  sp.fp.n = n;
  sp.fp.ldp = c->ld;
  sp.p_fej = (const double*)(d + o_pf);
  sp.lm_id = (const int*)(d + o_lm);
  sp.plane_sid = (const int*)(d + o_ps);
  sp.cp = (const double*)(d + o_cp);
  sp.cp_fej = (const double*)(d + o_cpf);
  sp.white_c = 1.0 / o->sigma_constraint;
  if (any_pre) {
    sp.pre_rows = (const int*)(d + o_pr);
    sp.pre_cols = (const int*)(d + o_pc);
    sp.pre_off = (const int*)(d + o_po);
    sp.pre_ids_off = (const int*)(d + o_pio);
    sp.pre_H = (const double*)(d + o_preH);
    sp.pre_ids = (const int*)(d + o_pid);
  }
  sp.row0 = (const int*)(d + o_r0);
  sp.gpos = (const int*)(d + o_gp);
  sp.Ht = c->Hd;
  sp.m_total = m_total;
  sp.gcols = gcols;
  sp.res_out = c->resd;
  sp.Hscr = c->slam_hscr;
  sp.rows_max = rows_max;
  sp.cols_max = cols_max;
  sp.h_in_lds = h_in_lds;
  const char* form_env = getenv("OVP_EKF_INFO_FORM");
  const bool info_form_only = form_env && form_env[0] == '1';
  // S-form (k_init.hip) up to 80 stacked rows: scratch [res 4 | dx n_max | 8 | chi2 L | status L] M_all | Linv | y in smallbuf, so
  // that everything the host wants comes back in ONE copy
  const bool sform = !info_form_only && m_total <= ovp_init_max_rows() && ovp_init_core_lds(0, m_total, gcols) <= ovp_init_max_lds();
  double* dres = c->smallbuf;
  double* dM = (double*)((char*)(dres + res_doubles) + lres_bytes);
  double* dLi = dM + (size_t)n * m_total;
  double* dy = dLi + (size_t)m_total * m_total;
  const bool sform_fits = (size_t)(dy + m_total + 8 - c->smallbuf) <= c->small_cap;
  if (sform && sform_fits) {
    sp.chi2 = dres + res_doubles;
    sp.status = (unsigned char*)(dres + res_doubles) + al(sizeof(double) * L);
    sp.Mall = dM;
  } else {
    sp.chi2 = (double*)c->slam_res;
    sp.status = (unsigned char*)c->slam_res + al(sizeof(double) * L);
  }
  HIPCHK(ovp_launch_slam_gate(&sp, L, ovp_slam_gate_lds(rows_max, cols_max, h_in_lds), s));
  const int* dgid = (const int*)(d + o_gi);
  char* hres = (char*)c->pl_hres;
  double* hres_d = (double*)hres;
  char* hl = hres + res_doubles * sizeof(double);  // [chi2 L | status L]
  auto finish_landmarks = [&]() {
    if (chi2_host) memcpy(chi2_host, hl, sizeof(double) * L);
    if (status_host) memcpy(status_host, hl + al(sizeof(double) * L), L);
    if (info) {
      info->n_cols = gcols;
      for (int l = 0; l < L; ++l) {
        const unsigned char st = ((unsigned char*)(hl + al(sizeof(double) * L)))[l];
        if (!st) continue;
        info->n_accepted++;
        const int rows_l = (l + 1 < L ? row0[l + 1] : m_total) - row0[l];
        const bool pre = any_pre && b->pre_rows[l] > 0;
        info->n_rows += (st == 2 && !pre) ? 2 * b->n_meas[l] : rows_l;
      }
    }
  };
  if (sform && sform_fits) {
    const int rows = m_total;
    HIPCHK(ovp_launch_init_core(c->P, c->ld, n, dgid, gcols, c->Hd, 0, rows, dM, c->resd /* unused: k = 0 */, c->resd, c->resd, 1.0, 1e300,
                                dLi, dy, dres, s));
    HIPCHK(ovp_launch_init_update(c->P, c->P_tmp, c->ld, n, dM, rows, 0, rows, dLi, dy, dres, dres + 4, s));
    HIPCHK(hipMemcpyAsync(hres, dres, res_doubles * sizeof(double) + lres_bytes, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    finish_landmarks();
    if (info) {
      info->not_spd = hres_d[1] > 0.5 ? 0 : 1;
      info->neg_diag = hres_d[2] != 0.0;
    }
    if (!(hres_d[1] > 0.5)) return OVP_E_NOTSPD;  // S = H P H^T + I lost definiteness: P is not a covariance; nothing was written
    double* t = c->P;
    c->P = c->P_tmp;
    c->P_tmp = t;
    if (dx_host) memcpy(dx_host, hres_d + 4, sizeof(double) * n);
    return hres_d[2] != 0.0 ? OVP_E_NEGDIAG : 0;
  }
  // information form: A = H^T H, b = H^T r on the call's columns, scattered to the state
  if (!c->Acc) HIPCHK(dalloc(&c->Acc, (size_t)c->n_max * c->n_max));
  if (!c->bcc) HIPCHK(dalloc(&c->bcc, (size_t)c->n_max));
  HIPCHK(hipMemsetAsync(c->flags, 0, sizeof(int) * 4, s));
  HIPCHK(ovp_launch_gemm(0, 1, gcols, gcols, m_total, c->Hd, m_total, c->Hd, m_total, c->Acc, gcols, 0, s));
  HIPCHK(ovp_launch_gemm(0, 0, gcols, 1, m_total, c->Hd, m_total, c->resd, 1, c->bcc, 1, 0, s));
  HIPCHK(hipMemsetAsync(c->Ab, 0, sizeof(double) * (size_t)(n + 1) * c->ld, s));
  HIPCHK(ovp_launch_scatter_gram(c->Acc, c->bcc, gcols, dgid, c->Ab, c->ld, n, s));
  {
    std::vector<int> ids(gids);
    std::sort(ids.begin(), ids.end());
    int rs = set_substate(c, ids);
    if (rs) return rs;
  }
  rc = ekf_from_gram(c, false);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(c->h_dx, c->dx, sizeof(double) * n, hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(c->h_flags, c->flags, sizeof(int) * 4, hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(hl, c->slam_res, lres_bytes, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  finish_landmarks();
  if (c->h_flags[0]) {  // positive semi-definite prior: S-form instead of the factor of P
    int rs = ekf_sform(c);
    if (rs) return rs;
  }
  if (dx_host) memcpy(dx_host, c->h_dx, sizeof(double) * n);
  if (info) {
    info->not_spd = c->h_flags[0];
    info->neg_diag = c->h_flags[1];
  }
  if (c->h_flags[0]) return OVP_E_NOTSPD;
  if (c->h_flags[1]) return OVP_E_NEGDIAG;
  return 0;
}

// ---- UpdaterSLAM::delayed_init, candidate loop on the device (update/UpdaterSLAM.cpp:204-364; csrc/k_dinit.hip) -----------------
extern "C" int ovp_slam_delayed_init(ovp_ctx* c, const ovp_update_opts* o, const ovp_feature_batch* b, uint8_t* ok_host,
                                     double* chi2_host, int* new_id, double* delta_init, double* dx_host, int dx_stride) {
  drop_kept_factor(c);  // (writes the covariance: a kept factor no longer belongs to it)
  if (!c || !o || !b || b->n_feats < 0) return OVP_E_ARG;
  if (!c->have_state || !c->have_cov) return OVP_E_STATE;
  const int L = b->n_feats, M = b->max_meas, n0 = c->n, ld = c->ld;
  if (L == 0) return 0;
  if (M < 2 || M > OVP_MAX_MEAS || !b->uv || !b->clone_idx || !b->n_meas || !b->p_FinG) return OVP_E_ARG;
  if (dx_host && dx_stride < n0 + 3 * L) return OVP_E_ARG;
  if (n0 + 3 * L > c->n_max) return OVP_E_CAPACITY;
  const unsigned calmask = (o->do_calib_camera_pose ? 0x3Fu : 0u) | (o->do_calib_camera_intrinsics ? (0xFFu << 6) : 0u);
  const int ncal = __builtin_popcount(calmask);
  int calcol[14];
  for (int k = 0; k < 14; ++k) {
    calcol[k] = (k < 6) ? c->calib_id + k : c->intr_id + (k - 6);
    if (!((calmask >> k) & 1)) calcol[k] = 0;
    else if (calcol[k] < 0 || calcol[k] >= n0) return OVP_E_ARG;
  }
  const int C = (int)c->h_clone_id.size();
  int cols_max = 1, rows_max = 4;
  for (int l = 0; l < L; ++l) {
    const int m = b->n_meas[l];
    if (m < 2 || m > M) return OVP_E_ARG;  // (update/UpdaterSLAM.cpp:112-118: the caller drops shorter tracks)
    for (int a = 0; a < m; ++a) {
      const int ci = b->clone_idx[(size_t)l * M + a];
      if (ci < 0 || ci >= C) return OVP_E_ARG;
    }
    const int cols = 6 * m + ncal, rup = 2 * m - 3;
    // outside the one-workgroup S-form (k_init.hip): the caller takes StateHelper::initialize candidate by candidate; nothing touched
    if (rup > ovp_init_max_rows() || ovp_init_core_lds(3, rup, cols) > ovp_init_max_lds() ||
        ovp_dinit_rows_lds(m, ncal) > OVP_DINIT_DYN_LDS) return OVP_E_CAPACITY;
    cols_max = std::max(cols_max, cols);
    rows_max = std::max(rows_max, 2 * m);
  }
  hipStream_t s = c->stream;
  auto al = [](size_t v) { return (v + 63) & ~(size_t)63; };
  // staging: the candidates as a feature batch + their column lists
  size_t off = 0;
  auto take = [&](size_t bytes) {
    const size_t o0 = off;
    off = al(off + bytes);
    return o0;
  };
  const size_t o_p = take(sizeof(double) * 3 * L), o_uv = take(sizeof(float) * 2 * (size_t)L * M), o_ci = take(sizeof(int) * (size_t)L * M),
               o_nm = take(sizeof(int) * L), o_id = take(sizeof(int) * (size_t)L * cols_max);
  const size_t stage_bytes = off;
  const size_t res_doubles = 4 + (size_t)c->n_max + 8;
  int rc = plane2_buffers(c, 0, stage_bytes, sizeof(double) * res_doubles * L + 64);
  if (rc) return rc;
  char* h = (char*)c->pl_hstage;
  char* d = (char*)c->pl_dstage;
  memcpy(h + o_p, b->p_FinG, sizeof(double) * 3 * L);
  memcpy(h + o_uv, b->uv, sizeof(float) * 2 * (size_t)L * M);
  memcpy(h + o_ci, b->clone_idx, sizeof(int) * (size_t)L * M);
  memcpy(h + o_nm, b->n_meas, sizeof(int) * L);
  for (int l = 0; l < L; ++l) {
    int* ids = (int*)(h + o_id) + (size_t)l * cols_max;
    const int m = b->n_meas[l];
    for (int a = 0; a < m; ++a)
      for (int k = 0; k < 6; ++k) ids[6 * a + k] = c->h_clone_id[b->clone_idx[(size_t)l * M + a]] + k;
    int q = 6 * m;
    for (int k = 0; k < 14; ++k)
      if ((calmask >> k) & 1) ids[q++] = calcol[k];
  }
  // device scratch: [result blocks L x res_doubles | Ht | Mall | Linv | y | Hinv 9 | Rk 9 | resid]
  const size_t n_end = (size_t)n0 + 3 * L;
  const size_t need = res_doubles * L + (size_t)cols_max * rows_max + n_end * rows_max + (size_t)rows_max * rows_max + rows_max + 32 +
                      rows_max + 64;
  if (need > c->dinit_cap) {
    if (c->dinit_buf) hipFree(c->dinit_buf);
    c->dinit_buf = nullptr;
    c->dinit_cap = 0;
    HIPCHK(dalloc(&c->dinit_buf, need + 1024));
    c->dinit_cap = need + 1024;
  }
  double* dres0 = c->dinit_buf;
  double* dHt = dres0 + res_doubles * L;
  double* dM = dHt + (size_t)cols_max * rows_max;
  double* dLi = dM + n_end * rows_max;
  double* dy = dLi + (size_t)rows_max * rows_max;
  double* dHinv = dy + rows_max + 8;
  double* dRk = dHinv + 12;
  double* dresid = dRk + 12;
  HIPCHK(hipMemcpyAsync(d, h, stage_bytes, hipMemcpyHostToDevice, s));
  ovp::DinitParams dp;
  memset(&dp, 0, sizeof(dp));
  dp.fp = c->fp;
  dp.fp.uv = (const float*)(d + o_uv);
  dp.fp.clone_idx = (const int*)(d + o_ci);
  dp.fp.n_meas = (const int*)(d + o_nm);
  dp.fp.p_FinG = (const double*)(d + o_p);
  dp.fp.n_feats = L;
  dp.fp.max_meas = M;
  dp.fp.do_fej = o->do_fej;
  dp.fp.calmask = calmask;
  for (int k = 0; k < 14; ++k) dp.fp.calcol[k] = calcol[k];
  dp.fp.white_px = 1.0 / o->sigma_px;
  dp.fp.ldp = ld;
  dp.n_max = c->n_max;
  dp.P = c->P;
  dp.clone_R = c->clone_R;
  dp.clone_p = c->clone_p;
  dp.cal = c->cal;
  dp.Ht = dHt;
  dp.Hinv = dHinv;
  dp.Rk = dRk;
  dp.resid = dresid;
  for (int l = 0; l < L; ++l) {
    const int m = b->n_meas[l], cols = 6 * m + ncal, rows = 2 * m, rup = rows - 3, n = n0 + 3 * l;
    dp.cand = l;
    dp.m_obs = m;
    dp.n = n;
    dp.prev_res = l ? dres0 + res_doubles * (l - 1) : nullptr;
    dp.ids = (const int*)(d + o_id) + (size_t)l * cols_max;
    memcpy(dp.idv, (const int*)(h + o_id) + (size_t)l * cols_max, sizeof(int) * cols);
    dp.res = dres0 + res_doubles * l;
    HIPCHK(ovp_launch_dinit_rows(&dp, ovp_dinit_rows_lds(m, ncal), s));
    HIPCHK(ovp_launch_init_m(c->P, ld, n, dp.ids, cols, dHt, rows, dM, s));  // M = P[:, ids] H_all^T on many workgroups
    // chi2 of the update rows with dof = all rows (StateHelper.cpp:471), initialize_invertible, update in place
    const double thr = o->chi2_multiplier * ovp_chi2_quantile_095(rows);
    HIPCHK(ovp_launch_init_core(c->P, ld, n, dp.ids, cols, dHt, 3, rup, dM, dHinv, dRk, dresid, 1.0, thr, dLi, dy, dp.res, s));
    HIPCHK(ovp_launch_init_update(c->P, c->P, ld, n + 3, dM, rows, 3, rup, dLi, dy, dp.res, dp.res + 4, s));
  }
  dp.cand = -1;
  dp.n = (int)n_end;
  dp.prev_res = dres0 + res_doubles * (L - 1);
  HIPCHK(ovp_launch_dinit_rows(&dp, 64, s));
  double* hres = (double*)c->pl_hres;
  HIPCHK(hipMemcpyAsync(hres, dres0, sizeof(double) * res_doubles * L, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  c->n = (int)n_end;
  // final layout: the inert blocks of the rejected candidates go (last first), the accepted ones move up
  std::vector<int> final_id(L, -1);
  int n_acc = 0, negdiag = 0;
  for (int l = 0; l < L; ++l) {
    const double* r = hres + res_doubles * l;
    if (r[1] > 0.5) final_id[l] = n0 + 3 * n_acc++;
    if (r[1] > 0.5 && r[2] != 0.0) negdiag = 1;
  }
  for (int l = L - 1; l >= 0; --l)
    if (final_id[l] < 0) {
      rc = ovp_cov_marginalize(c, n0 + 3 * l, 3);
      if (rc) return rc;
    }
  for (int l = 0; l < L; ++l) {
    const double* r = hres + res_doubles * l;
    const bool ok = r[1] > 0.5;
    if (ok_host) ok_host[l] = ok ? 1 : 0;
    if (chi2_host) chi2_host[l] = r[0];
    if (new_id) new_id[l] = final_id[l];
    if (delta_init)
      for (int k = 0; k < 3; ++k) delta_init[3 * l + k] = ok ? r[4 + c->n_max + k] : 0.0;
    if (dx_host) {
      double* dx = dx_host + (size_t)l * dx_stride;
      memset(dx, 0, sizeof(double) * dx_stride);
      if (ok) {
        memcpy(dx, r + 4, sizeof(double) * n0);
        for (int g = 0; g <= l; ++g)  // the landmarks that were state variables at that point, at their final ids
          if (final_id[g] >= 0)
            for (int k = 0; k < 3; ++k) dx[final_id[g] + k] = r[4 + n0 + 3 * g + k];
      }
    }
  }
  return negdiag ? OVP_E_NEGDIAG : 0;
}

// ---- propagation / clone / marginalise ---------------------------------------------------------
extern "C" int ovp_cov_propagate(ovp_ctx* c, int new_start, int phi_size, const int* old_ids, const int* old_sizes,
                                 int n_old, const double* Phi_host, const double* Q_host, int* neg_diag) {
  drop_kept_factor(c);  // (writes the covariance: a kept factor no longer belongs to it)
  if (!c || !old_ids || !old_sizes || !Phi_host || !Q_host || phi_size < 1 || n_old < 1) return OVP_E_ARG;
  if (!c->have_cov) return OVP_E_STATE;
  const int n = c->n;
  if (new_start < 0 || new_start + phi_size > n || phi_size > 64) return OVP_E_ARG;
  std::vector<int> oldcol;
  for (int i = 0; i < n_old; ++i)
    for (int k = 0; k < old_sizes[i]; ++k) {
      if (old_ids[i] < 0 || old_ids[i] + k >= n) return OVP_E_ARG;
      oldcol.push_back(old_ids[i] + k);
    }
  const int nold = (int)oldcol.size();
  if (nold > 4 * c->n_max) return OVP_E_CAPACITY;
  double* dCPT = c->smallbuf;
  double* dPCP = dCPT + (size_t)n * phi_size;
  if ((size_t)(dPCP + (size_t)phi_size * phi_size - c->smallbuf) > c->small_cap) return OVP_E_CAPACITY;
  HIPCHK(hipMemsetAsync(c->flags, 0, sizeof(int) * 4, c->stream));
  {
    // [Phi | Q | ids] packed into the pinned arena, one copy; the kernels read them from the device half of the arena
    void *ah = nullptr, *ad = nullptr;
    const size_t b_pq = sizeof(double) * ((size_t)phi_size * nold + (size_t)phi_size * phi_size);
    const size_t o_id = ((b_pq + 63) / 64) * 64, bytes = o_id + sizeof(int) * (size_t)nold;
    const int rca = ovp_io_arena(c, bytes, &ah, &ad);
    if (rca) return rca;
    memcpy(ah, Phi_host, sizeof(double) * phi_size * nold);
    memcpy((double*)ah + (size_t)phi_size * nold, Q_host, sizeof(double) * phi_size * phi_size);
    memcpy((char*)ah + o_id, oldcol.data(), sizeof(int) * nold);
    HIPCHK(hipMemcpyAsync(ad, ah, bytes, hipMemcpyHostToDevice, c->stream));
    const double* dPhi = (const double*)ad;
    const double* dQ = dPhi + (size_t)phi_size * nold;
    HIPCHK(ovp_launch_propagate(c->P, c->ld, n, new_start, phi_size, (const int*)((char*)ad + o_id), nold, dPhi, dQ, dCPT, dPCP,
                                c->flags + 1, c->stream));
  }
  HIPCHK(hipMemcpyAsync(c->h_flags, c->flags, sizeof(int) * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (neg_diag) *neg_diag = c->h_flags[1];
  return c->h_flags[1] ? OVP_E_NEGDIAG : 0;
}

extern "C" int ovp_cov_clone(ovp_ctx* c, int src_id, int size) {
  drop_kept_factor(c);  // (writes the covariance: a kept factor no longer belongs to it)
  if (!c || size < 1 || src_id < 0) return OVP_E_ARG;
  if (!c->have_cov) return OVP_E_STATE;
  if (src_id + size > c->n) return OVP_E_ARG;
  if (c->n + size > c->n_max) return OVP_E_CAPACITY;
  HIPCHK(ovp_launch_cov_clone(c->P, c->ld, c->n, src_id, size, c->clone_jitter, c->stream));
  c->n += size;
  return 0;
}

extern "C" int ovp_cov_clone_jitter(ovp_ctx* c, double relative_inflation) {
  if (!c || !(relative_inflation >= 0.0) || relative_inflation > 1e-6) return OVP_E_ARG;
  c->clone_jitter = relative_inflation;
  return 0;
}

extern "C" int ovp_cov_marginalize(ovp_ctx* c, int id, int size) {
  drop_kept_factor(c);  // (writes the covariance: a kept factor no longer belongs to it)
  if (!c || size < 1 || id < 0) return OVP_E_ARG;
  if (!c->have_cov) return OVP_E_STATE;
  if (id + size > c->n) return OVP_E_ARG;
  HIPCHK(ovp_launch_cov_marginalize(c->P, c->P_tmp, c->ld, c->n, id, size, c->stream));
  double* t = c->P;
  c->P = c->P_tmp;
  c->P_tmp = t;
  c->n -= size;
  return 0;
}

extern "C" int ovp_cov_initialize_invertible(ovp_ctx* c, const double* H_R, int k, int cols, int ld, const int* col_ids,
                                             const double* H_Linv, const double* R) {
  drop_kept_factor(c);  // (writes the covariance: a kept factor no longer belongs to it)
  if (!c || !H_R || !col_ids || !H_Linv || !R || k < 1 || k > 6 || cols < 1 || ld < k) return OVP_E_ARG;
  if (!c->have_cov) return OVP_E_STATE;
  const int n = c->n;
  if (n + k > c->n_max) return OVP_E_CAPACITY;
  if (cols > c->n_max) return OVP_E_ARG;
  for (int j = 0; j < cols; ++j)
    if (col_ids[j] < 0 || col_ids[j] >= n) return OVP_E_ARG;
  // device layout: H_R row-major [k][cols], Hinv / R row-major [k][k], M_a [n][6]
  std::vector<double> hr((size_t)k * cols), hi((size_t)k * k), rk((size_t)k * k);
  for (int i = 0; i < k; ++i)
    for (int a = 0; a < cols; ++a) hr[(size_t)i * cols + a] = H_R[(size_t)a * ld + i];
  for (int i = 0; i < k; ++i)
    for (int j = 0; j < k; ++j) {
      hi[(size_t)i * k + j] = H_Linv[(size_t)j * k + i];
      rk[(size_t)i * k + j] = R[(size_t)j * k + i];
    }
  double* dHR = c->smallbuf;
  double* dHi = dHR + (size_t)k * cols;
  double* dRk = dHi + 36;
  double* dMa = dRk + 36;
  if ((size_t)(dMa + (size_t)6 * n - c->smallbuf) > c->small_cap) return OVP_E_CAPACITY;
  HIPCHK(hipMemcpyAsync(dHR, hr.data(), sizeof(double) * hr.size(), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(dHi, hi.data(), sizeof(double) * hi.size(), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(dRk, rk.data(), sizeof(double) * rk.size(), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(c->idbuf, col_ids, sizeof(int) * cols, hipMemcpyHostToDevice, c->stream));
  HIPCHK(ovp_launch_init_invertible(c->P, c->ld, n, c->idbuf, cols, dHR, k, dMa, dHi, dRk, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));  // host vectors above must outlive the copies
  c->n = n + k;
  return 0;
}

extern "C" int ovp_cov_augment_dt(ovp_ctx* c, int pose_id, int dt_id, const double dnc_dt[6]) {
  drop_kept_factor(c);  // (writes the covariance: a kept factor no longer belongs to it)
  if (!c || !dnc_dt) return OVP_E_ARG;
  if (!c->have_cov) return OVP_E_STATE;
  if (pose_id < 0 || pose_id + 6 > c->n || dt_id < 0 || dt_id >= c->n) return OVP_E_ARG;
  HIPCHK(ovp_launch_augment_dt(c->P, c->ld, c->n, pose_id, dt_id, dnc_dt, c->stream));
  return 0;
}

// ---- diagnostics -------------------------------------------------------------------------------
// ---- StateHelper::initialize as one device sequence (csrc/k_init.hip) ---------------------------------------------------
extern "C" int ovp_cov_initialize(ovp_ctx* c, const double* Hx_init, const double* H_up, int k, int rup, int cols, const int* col_ids,
                                  const double* H_Linv, const double* R_init, const double* res_up, double r_iso,
                                  double chi2_threshold, int do_update, int* accepted, double* chi2, double* dx_host) {
  drop_kept_factor(c);  // (writes the covariance: a kept factor no longer belongs to it)
  if (!c || !Hx_init || !col_ids || !H_Linv || !R_init || k < 1 || k > 6 || cols < 1 || rup < 0) return OVP_E_ARG;
  if (rup > 0 && (!H_up || !res_up || !(r_iso > 0.0))) return OVP_E_ARG;
  if (!c->have_cov) return OVP_E_STATE;
  const int n = c->n, n2 = n + k, ld = c->ld, m = k + rup;
  if (n2 > c->n_max || cols > c->n_max) return OVP_E_CAPACITY;
  // S = H P H^T + R and the gathered rows of P H^T live in the LDS of one workgroup: the caller takes the three separate calls
  // for more rows than that holds (OVP_E_CAPACITY, nothing has been touched)
  if (rup > ovp_init_max_rows() || ovp_init_core_lds(k, rup, cols) > ovp_init_max_lds()) return OVP_E_CAPACITY;
  for (int j = 0; j < cols; ++j)
    if (col_ids[j] < 0 || col_ids[j] >= n) return OVP_E_ARG;
  hipStream_t s = c->stream;
  const bool upd = rup > 0 && do_update;
  // one pinned staging block: [H_all^T cols x m | Hinv 36 | Rk 36 | res rup] ids
  const size_t oHt = 0, oHi = oHt + (size_t)cols * m, oRk = oHi + 36, oRes = oRk + 36, oId = oRes + rup + 8;
  const size_t bytes = oId * sizeof(double) + sizeof(int) * (size_t)cols + 64;
  const size_t res_doubles = 4 + (size_t)c->n_max + 8;
  int rc = plane2_buffers(c, 0, bytes, res_doubles * sizeof(double));  // the plane loop's pinned staging and result blocks
  if (rc) return rc;
  double* h = (double*)c->pl_hstage;
  double* d = (double*)c->pl_dstage;
  for (int a = 0; a < cols; ++a) {
    double* row = h + oHt + (size_t)a * m;
    for (int i = 0; i < k; ++i) row[i] = Hx_init[(size_t)a * k + i];
    for (int i = 0; i < rup; ++i) row[k + i] = H_up[(size_t)a * rup + i];
  }
  for (int i = 0; i < 36; ++i) h[oHi + i] = h[oRk + i] = 0.0;
  for (int i = 0; i < k; ++i)
    for (int j = 0; j < k; ++j) {
      h[oHi + (size_t)i * k + j] = H_Linv[(size_t)j * k + i];
      h[oRk + (size_t)i * k + j] = R_init[(size_t)j * k + i];
    }
  for (int i = 0; i < rup; ++i) h[oRes + i] = res_up[i];
  memcpy(h + oId, col_ids, sizeof(int) * cols);
  const int* did = (const int*)(d + oId);
  // device scratch: result block [chi2 | accept | negdiag | - | dx n2], M_all [n2 x m], Linv [rup x rup], y [rup]
  double* dres = c->smallbuf;
  double* dM = dres + res_doubles;
  double* dLi = dM + (size_t)n2 * m;
  double* dy = dLi + (size_t)rup * rup;
  if ((size_t)(dy + rup + 8 - c->smallbuf) > c->small_cap) return OVP_E_CAPACITY;
  HIPCHK(hipMemcpyAsync(c->pl_dstage, c->pl_hstage, bytes, hipMemcpyHostToDevice, s));
  HIPCHK(ovp_launch_init_m(c->P, ld, n, did, cols, d + oHt, m, dM, s));
  HIPCHK(ovp_launch_init_core(c->P, ld, n, did, cols, d + oHt, k, rup, dM, d + oHi, d + oRk, d + oRes, r_iso > 0.0 ? r_iso : 1.0,
                              chi2_threshold, dLi, dy, dres, s));
  double* hres = (double*)c->pl_hres;
  if (upd) {
    // P+ = P - W W^T goes to the second covariance buffer (a tile reads entries other tiles overwrite)
    HIPCHK(ovp_launch_init_update(c->P, c->P_tmp, ld, n2, dM, m, k, rup, dLi, dy, dres, dres + 4, s));
    HIPCHK(hipMemcpyAsync(hres, dres, sizeof(double) * (4 + (size_t)n2), hipMemcpyDeviceToHost, s));
  } else {
    HIPCHK(hipMemcpyAsync(hres, dres, sizeof(double) * 4, hipMemcpyDeviceToHost, s));
  }
  HIPCHK(hipStreamSynchronize(s));
  const bool ok = hres[1] > 0.5;
  if (accepted) *accepted = ok ? 1 : 0;
  if (chi2) *chi2 = hres[0];
  if (!ok) return 0;
  c->n = n2;
  if (upd) {
    double* t = c->P;
    c->P = c->P_tmp;
    c->P_tmp = t;
  }
  if (dx_host) {
    if (upd) memcpy(dx_host, hres + 4, sizeof(double) * n2);
    else memset(dx_host, 0, sizeof(double) * n2);
  }
  if (upd && hres[2] != 0.0) return OVP_E_NEGDIAG;
  return 0;
}

extern "C" long ovp_debug_read(ovp_ctx* c, const char* name, void* host, long max_bytes) {
  if (!c || !name || !host) return OVP_E_ARG;
  const size_t nn = (size_t)(c->n_max + 1) * c->ld * sizeof(double);
  const void* src = nullptr;
  size_t bytes = 0;
  if (!strcmp(name, "A") || !strcmp(name, "Ab")) { src = c->Ab; bytes = nn; }
  else if (!strcmp(name, "L")) { src = c->L; bytes = nn; }
  else if (!strcmp(name, "T")) { src = c->T; bytes = nn; }
  else if (!strcmp(name, "Lt")) { src = c->Lt; bytes = nn; }
  else if (!strcmp(name, "Y")) { src = c->Y; bytes = nn; }
  else if (!strcmp(name, "W1")) { src = c->W1; bytes = nn; }
  else if (!strcmp(name, "P")) { src = c->P; bytes = nn; }
  else if (!strcmp(name, "G")) { src = c->G; bytes = (size_t)3 * c->n_feats * c->ldg * sizeof(double); }
  else if (!strcmp(name, "rec")) { src = c->rec; bytes = (size_t)c->fp.n_clones * c->n_feats * 2 * 21 * sizeof(double); }
  else if (!strcmp(name, "plres")) { src = c->pl_res; bytes = (size_t)4 * c->pl_cap * sizeof(double); if (!src) return OVP_E_STATE; }
  else if (!strcmp(name, "An")) { src = c->pl_An; bytes = nn; if (!src) return OVP_E_STATE; }
  else if (!strcmp(name, "bn")) { src = c->pl_bn; bytes = (size_t)c->n_max * sizeof(double); if (!src) return OVP_E_STATE; }
  else if (!strcmp(name, "chi2")) {
    hipStreamSynchronize(c->stream);
    bytes = (size_t)c->n_feats * sizeof(double);
    if ((long)bytes > max_bytes) bytes = (size_t)max_bytes;
    memcpy(host, c->h_chi2, bytes);
    return (int)bytes;
  }
  else if (!strncmp(name, "bench_chol", 10)) {
    // diagnostics: average time of k_tilechol on the resident covariance; name = "bench_chol<skipmask>"
    ovp_dbg_tilechol_skip = atoi(name + 10);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) ovp_launch_tilechol(c->P, c->L, c->Dinv, c->Ltp, c->n, c->ld, c->flags + 3, 0, c->stream);
    hipEventRecord(e0, c->stream);
    for (int i = 0; i < 20; ++i) ovp_launch_tilechol(c->P, c->L, c->Dinv, c->Ltp, c->n, c->ld, c->flags + 3, 0, c->stream);
    hipEventRecord(e1, c->stream);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    ovp_dbg_tilechol_skip = 0;
    *(double*)host = ms / 20.0;
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return 8;
  }
  else if (!strcmp(name, "cycles_on")) {
    if (!c->dbg_cycles && hipMalloc((void**)&c->dbg_cycles, (size_t)c->f_max * 10 * sizeof(long long)) != hipSuccess) return OVP_E_STATE;
    return 0;
  }
  else if (!strcmp(name, "cycles")) { src = c->dbg_cycles; bytes = (size_t)c->n_feats * 10 * sizeof(long long); if (!src) return OVP_E_STATE; }
  else return OVP_E_ARG;
  if ((long)bytes > max_bytes) bytes = (size_t)max_bytes;
  if (hipStreamSynchronize(c->stream) != hipSuccess) return OVP_E_STATE;
  if (hipMemcpy(host, src, bytes, hipMemcpyDeviceToHost) != hipSuccess) return OVP_E_STATE;
  return (long)bytes;
}

// Diagnostics / micro-benchmark of the second-generation tile Cholesky (k_chol2.hip): factorizes the n x n host matrix A (+ I)
// bordered with the row brow, returns the dense factor of the bordered matrix ((n+1) x (n+1) row-major, or n x n without a border),
// z = L^-1 brow, y = L^-T z and the pivots; avg_ms = average duration of `reps` launches (HIP events).
static double g_dbg_chol2_floor = 0.0;
extern "C" void ovp_debug_chol2_floor(double piv_floor) { g_dbg_chol2_floor = piv_floor; }

extern "C" int ovp_debug_chol2(ovp_ctx* c, const double* A_host, int n, int lda, const double* brow_host, int add_identity,
                               double* L_host, double* z_host, double* y_host, double* piv_host, int reps, float* avg_ms) {
  if (!c || !A_host || n < 1 || lda < n) return OVP_E_ARG;
  const int nb = brow_host ? n + 1 : n;
  if (nb > ovp_chol2_max_n() + 1) return OVP_E_CAPACITY;
  double *dA = nullptr, *dL = nullptr, *dv = nullptr;
  HIPCHK(dalloc(&dA, (size_t)n * n));
  HIPCHK(dalloc(&dL, (size_t)nb * nb));
  HIPCHK(dalloc(&dv, (size_t)4 * n + 16));
  HIPCHK(hipMemcpy2D(dA, sizeof(double) * n, A_host, sizeof(double) * lda, sizeof(double) * n, n, hipMemcpyHostToDevice));
  if (brow_host) HIPCHK(hipMemcpy(dv, brow_host, sizeof(double) * n, hipMemcpyHostToDevice));
  HIPCHK(hipMemset(dL, 0, sizeof(double) * (size_t)nb * nb));
  HIPCHK(hipMemsetAsync(c->flags, 0, sizeof(int) * 4, c->stream));
  ovp::Chol2Job j;
  memset(&j, 0, sizeof(j));
  j.A = dA;
  j.n = n;
  j.ld = n;
  j.add_identity = add_identity;
  j.mode = 0;
  j.brow = brow_host ? dv : nullptr;
  j.flag = c->flags;
  j.Ldense = dL;
  j.ldo = nb;
  j.z_out = brow_host ? dv + n : nullptr;
  j.y_out = brow_host ? dv + 2 * n : nullptr;
  j.piv_out = dv + 3 * n;
  j.piv_floor = g_dbg_chol2_floor;
  HIPCHK(ovp_launch_chol2(&j, nullptr, nullptr, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (getenv("OVP_C2_STAMPS")) {
    long long* st = nullptr;
    HIPCHK(hipMalloc((void**)&st, sizeof(long long) * 16 * 32));
    HIPCHK(hipMemset(st, 0, sizeof(long long) * 16 * 32));
    ovp::Chol2Job jt = j;
    jt.Ldense = nullptr;
    if (!getenv("OVP_C2_TIME_Y")) jt.y_out = nullptr;
    jt.stamps = st;
    HIPCHK(ovp_launch_chol2(&jt, nullptr, nullptr, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    long long h[16 * 32];
    HIPCHK(hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost));
    hipFree(st);
    const int nt = (nb + 15) / 16;
    fprintf(stderr, "chol2 stamps (cycles): elimination wave 0 [wait column | read+eliminate | write] ; tile wave 0 [wait panel | reload + next column | wait buffer + publish | rest]\n");
    fprintf(stderr, " tile wave 0 prologue: issue loads %lld, patch special tiles %lld, publish column 0 %lld (elimination wave 0 starts waiting at %lld after the tile wave)\n",
            h[13] - h[16 + 13], h[14] - h[13], h[15] - h[14], h[0] - h[16 + 13]);
    for (int k = 0; k < nt; ++k) {
      const long long* e = h + k * 16;
      fprintf(stderr, " k=%2d E: %6lld %6lld %6lld | T: %6lld %6lld %6lld %6lld | E step %6lld T step %6lld | on arrival: column %+lld panel %+lld trail %+lld | E start %lld T start %lld\n", k, e[1] - e[0], e[2] - e[1],
              e[3] - e[2], e[9] - e[8], e[10] - e[9], e[11] - e[10], e[12] - e[11], e[3] - e[0], e[12] - e[8], e[4] / 1000000 - 500, (e[4] / 1000) % 1000 - 500, e[4] % 1000 - 500, e[0] - h[0], e[8] - h[0]);
    }
    if (jt.y_out) {
      fprintf(stderr, "back substitution: preparation (sub-diagonal tiles to LDS, inverses of the diagonal blocks) %lld cycles + barrier %lld, chain %lld\n",
              h[nt * 16 + 4] - h[nt * 16 + 3], h[nt * 16 + 5] - h[nt * 16 + 4], h[7] - h[nt * 16 + 5]);
      fprintf(stderr, "back substitution, wave 0 per step: [first product + loads | wait for the partial sums | sum, second product, publish]\n");
      for (int k = nt - 1; k >= 0; --k) {
        const long long* e = h + k * 16;
        fprintf(stderr, " k=%2d  %6lld %6lld %6lld | step %6lld\n", k, e[5] - e[4], k <= nt - 3 ? e[6] - e[5] : 0LL,
                e[7] - (k <= nt - 3 ? e[6] : e[5]), e[7] - e[4]);
      }
    }
  }
  if (reps > 0) {
    ovp::Chol2Job jt = j;  // timing: the factorization alone (no dense output)
    jt.Ldense = nullptr;
    if (!getenv("OVP_C2_TIME_Y")) jt.y_out = nullptr;  // OVP_C2_TIME_Y: with the back substitution
    jt.dbg = getenv("OVP_C2_DBG") ? atoi(getenv("OVP_C2_DBG")) : 0;
    HIPCHK(hipEventRecord(c->ev_t[0], c->stream));
    for (int r = 0; r < reps; ++r) HIPCHK(ovp_launch_chol2(&jt, nullptr, nullptr, c->stream));
    HIPCHK(hipEventRecord(c->ev_t[1], c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    float ms = 0.f;
    hipEventElapsedTime(&ms, c->ev_t[0], c->ev_t[1]);
    if (avg_ms) *avg_ms = ms / reps;
  }
  if (L_host) HIPCHK(hipMemcpy(L_host, dL, sizeof(double) * (size_t)nb * nb, hipMemcpyDeviceToHost));
  if (z_host && brow_host) HIPCHK(hipMemcpy(z_host, dv + n, sizeof(double) * n, hipMemcpyDeviceToHost));
  if (y_host && brow_host) HIPCHK(hipMemcpy(y_host, dv + 2 * n, sizeof(double) * n, hipMemcpyDeviceToHost));
  if (piv_host) HIPCHK(hipMemcpy(piv_host, dv + 3 * n, sizeof(double) * n, hipMemcpyDeviceToHost));
  int fl[4];
  HIPCHK(hipMemcpy(fl, c->flags, sizeof(fl), hipMemcpyDeviceToHost));
  HIPCHK(hipMemsetAsync(c->flags, 0, sizeof(int) * 4, c->stream));
  hipFree(dA);
  hipFree(dL);
  hipFree(dv);
  return (fl[0] & 2) ? OVP_E_TIMEOUT : (fl[0] ? OVP_E_NOTSPD : 0);
}

extern "C" int ovp_last_timings(ovp_ctx* c, float* ms4) {
  if (!c || !ms4) return OVP_E_ARG;
  memcpy(ms4, c->last_ms, sizeof(float) * 4);
  return 0;
}

extern "C" int ovp_plane_kernel_timer(ovp_ctx* c, int enable, int reset, float* avg_ms, int* n_launches) {
  if (!c) return OVP_E_ARG;
  if (avg_ms) *avg_ms = c->pl_klaunches ? (float)(c->pl_ktime_ms / c->pl_klaunches) : 0.f;
  if (n_launches) *n_launches = c->pl_klaunches;
  if (reset) {
    c->pl_ktime_ms = 0.0;
    c->pl_klaunches = 0;
  }
  c->pl_ktimer = enable;
  return 0;
}

extern "C" int ovp_kernel_timer(ovp_ctx* c, int enable, int reset, float* avg_ms_feat, int* n_launches) {
  if (!c) return OVP_E_ARG;
  if (avg_ms_feat) *avg_ms_feat = c->klaunches ? (float)(c->ktime_ms / c->klaunches) : 0.f;
  if (n_launches) *n_launches = c->klaunches;
  if (reset) {
    c->ktime_ms = 0.0;
    c->klaunches = 0;
  }
  c->ktimer = enable != 0;
  return 0;
}
