/*
 * libovplane_hip.so - C-ABI of the MI355X-native MSCKF(+plane) EKF update path for rpng/ov_plane.
 *
 * The reference has no FFI layer: the boundary is the C++ surface that VioManager calls
 * (SURVEY.md §8b).  Each entry point below names the reference method(s) it replaces; citations are
 * relative to /root/reference/ov_plane/src/.  Plain pointers and sizes only, no C++/torch types.
 *
 * Conventions
 *   - all matrices are f64; the covariance is handed over ROW-major == column-major (it is symmetric)
 *     with an explicit leading dimension;
 *   - `id` always means Type::id() = column offset of a variable in State::_Cov;
 *   - every function returns 0 on success, a positive hipError_t, or a negative OVP_E_* code;
 *   - one context per filter, not thread-safe (matches the reference: at most one update in flight - the staged entry points keep
 *     per-update words in host-mapped memory that the kernels read, e.g. the output slots of the features of a point update, so
 *     ovp_msckf_fetch_results must have returned before the next update of the same context is enqueued);
 *   - work is enqueued on the context's stream; functions that return results to host memory
 *     synchronise that stream unless their name ends in _async (then call ovp_sync()).
 */
#ifndef OVPLANE_HIP_H
#define OVPLANE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OVP_MAX_MEAS 32 /* max observations per feature handled by the wave-per-feature kernels */

enum {
  OVP_E_ARG = -1,       /* bad argument */
  OVP_E_CAPACITY = -2,  /* exceeds the capacity given to ovp_ctx_create */
  OVP_E_NOTSPD = -3,    /* a Cholesky factorisation hit a non-positive pivot */
  OVP_E_NEGDIAG = -4,   /* negative covariance diagonal (reference: std::exit, StateHelper.cpp:177-187) */
  OVP_E_NODEVICE = -5,  /* no HIP device / wrong architecture */
  OVP_E_STATE = -6,     /* call order violated (e.g. update before upload) */
  OVP_E_TIMEOUT = -7,   /* a device-side hand-over between the two workgroups of the plane solve did not arrive (bounded spin) */
  OVP_E_RCCL = -8,      /* librccl could not be loaded, or an RCCL call failed (message on stderr) */
  OVP_E_PEER = -9       /* sharded update: another rank's build failed; every rank of the call reports an error (errors are collective) */
};

typedef struct ovp_ctx ovp_ctx;

/* Options read on the path: update/UpdaterOptions.h:37-53, state/StateOptions.h:41-153. */
typedef struct {
  double sigma_px;                /* UpdaterOptions::sigma_pix                 */
  double chi2_multiplier;         /* UpdaterOptions::chi2_multipler            */
  double sigma_constraint;        /* StateOptions::sigma_constraint            */
  int do_fej;                     /* StateOptions::do_fej                      */
  int do_calib_camera_pose;       /* StateOptions::do_calib_camera_pose        */
  int do_calib_camera_intrinsics; /* StateOptions::do_calib_camera_intrinsics  */
  int skip_plane_used;            /* point update only: 1 = leave out the features the accepted planes of the preceding
                                     ovp_msckf_plane_update on this batch consumed (the reference erases them from feature_vec
                                     before the point loop, update/UpdaterMSCKF.cpp:657-666); the mask stays on the device */
} ovp_update_opts;

/* Values of the ov_type variables the Jacobians read (state/State.h:86-121): clone poses (value and
 * first-estimate), camera extrinsics/intrinsics.  MONOCULAR ONLY: one camera (cam 0, radtan or equidistant) - every shipped
 * configuration of the reference sets max_cameras: 1; the per-camera measurement loop of update/UpdaterHelper.cpp:335-344 is
 * taken for a single camera id, a stereo pair would need a second calibration block and camera index per measurement.
 * Host pointers. */
typedef struct {
  int n_state;               /* N = State::_Cov.rows()                       */
  int n_clones;
  const double *clone_q;     /* [n_clones*4] JPL q_GtoI: PoseJPL::Rot()      */
  const double *clone_p;     /* [n_clones*3] p_IinG:     PoseJPL::pos()      */
  const double *clone_q_fej; /* PoseJPL::Rot_fej()                           */
  const double *clone_p_fej; /* PoseJPL::pos_fej()                           */
  const int *clone_id;       /* [n_clones] Type::id()                        */
  double calib_q[4];         /* R_ItoC (State::_calib_IMUtoCAM.at(0))        */
  double calib_p[3];         /* p_IinC                                       */
  int calib_id;              /* ignored unless do_calib_camera_pose          */
  double intrinsics[8];      /* fx fy cx cy k1 k2 p1 p2 (State::_cam_intrinsics.at(0)); k1..k4 for the fisheye model */
  int intr_id;               /* ignored unless do_calib_camera_intrinsics    */
  int cam_fisheye;           /* 0 = ext CamRadtan, 1 = ext CamEqui (State::_cam_intrinsics_cameras.at(0), call sites
                                update/UpdaterHelper.cpp:365,389) */
} ovp_state_tables;

/* SoA form of a vector of UpdaterHelper::UpdaterHelperFeature (update/UpdaterHelper.h:62-105),
 * GLOBAL_3D representation (the only one the shipped configs use; UpdaterHelper.cpp:39-43,455-456). */
typedef struct {
  int n_feats;
  int max_meas;          /* row pitch of uv / clone_idx, <= OVP_MAX_MEAS                       */
  const float *uv;       /* [n_feats*max_meas*2] raw pixel measurements (f32, UpdaterHelper.h:68) */
  const int *clone_idx;  /* [n_feats*max_meas] index into the clone tables, -1 = padding       */
  const int *n_meas;     /* [n_feats]                                                          */
  const double *p_FinG;  /* [n_feats*3] linearisation point (fej == value for MSCKF features)  */
} ovp_feature_batch;

/* Result summary of one update step. */
typedef struct {
  int n_accepted;      /* features that passed the chi2 gate (UpdaterMSCKF.cpp:755)            */
  int n_rows;          /* stacked rows before compression = sum (2m-3) over accepted features   */
  int n_cols;          /* involved state columns                                                */
  int neg_diag;        /* 1 if the updated covariance has a negative diagonal entry             */
  int not_spd;         /* 1 if a factorisation failed                                           */
  int reserved[3];
} ovp_update_info;

#define OVP_PLANE_MAX_SLAM 80 /* SLAM landmarks on ONE out-of-state plane handled by ovp_msckf_plane_update: above every
                                 max_slam_features of the shipped configurations (25 .. 75), so that no plane of a real frame meets it;
                                 each landmark also adds three involved columns to the loop's 287-column budget */

/* ---- context --------------------------------------------------------------------------------- */
/* stream: a hipStream_t (e.g. torch.cuda.current_stream().cuda_stream) or NULL to let the library create its own.
 * Sizes: n_state_max <= 700 (the feature kernels stage projector rows in LDS); states up to 288 columns take the register-
 * resident factorizations, larger ones the sub-state update (a batch may then touch at most 288 columns to stay fast);
 * ovp_msckf_plane_update above 287 columns runs its loop on the columns the planes of the call involve (clones, calibration, the
 * planes that are state variables, the SLAM landmarks on the others: at most 287 of them, OVP_E_CAPACITY beyond) and carries the
 * rest of the state along; ovp_plane_init runs on the marginal of the clone and calibration columns at any state size. */
int ovp_ctx_create(int device, int n_state_max, int n_clones_max, int n_feats_max, void *stream, ovp_ctx **out);
int ovp_ctx_destroy(ovp_ctx *ctx);
int ovp_sync(ovp_ctx *ctx);
/* the hipStream_t all work of the context is ordered on (to enqueue a collective between the staged calls) */
int ovp_ctx_stream(ovp_ctx *ctx, void **stream);
const char *ovp_version(void);
const char *ovp_error_string(int code);

/* ---- covariance residency (State::_Cov lives on the device) --------------------------------- */
/* replaces direct access to State::_Cov (friend StateHelper, state/State.h:123-133) */
int ovp_cov_upload(ovp_ctx *ctx, const double *P_host, int n, int ld);
int ovp_cov_download(ovp_ctx *ctx, double *P_host, int n, int ld);
/* device-to-device variant: P_dev is a device pointer (n x n, ld) */
int ovp_cov_set_device(ovp_ctx *ctx, const double *P_dev, int n, int ld);
/* StateHelper::get_marginal_covariance (state/StateHelper.cpp:231-259): out is host [sum(size)^2] col-major */
int ovp_cov_marginal(ovp_ctx *ctx, const int *ids, const int *sizes, int n_vars, double *out_host);

/* ---- state tables / feature batch ----------------------------------------------------------- */
int ovp_state_upload(ovp_ctx *ctx, const ovp_state_tables *st);
/* host -> device copy of a feature batch */
int ovp_batch_upload(ovp_ctx *ctx, const ovp_feature_batch *host_batch);
/* zero-copy: the pointers inside dev_batch are DEVICE pointers that stay valid until the next bind/upload */
int ovp_batch_bind_device(ovp_ctx *ctx, const ovp_feature_batch *dev_batch);
/* Restricts the POINT updates that follow (ovp_msckf_update / ovp_msckf_build_gate_gram_async) to the features [lo, hi) of the
 * uploaded batch - the shard of one rank when the whole frame is resident on every GPU (the plane loop runs on the
 * whole frame everywhere, the leftovers are split by index range, no second upload).  lo = hi = -1 lifts the restriction, lo >= hi
 * >= 0 is an empty shard; every upload / bind lifts it. */
int ovp_batch_set_range(ovp_ctx *ctx, int lo, int hi);

/* ---- the update step ------------------------------------------------------------------------ */
/* UpdaterMSCKF::update point-feature path (update/UpdaterMSCKF.cpp:671-814):
 *   get_feature_jacobian_full (UpdaterHelper.cpp:195-513) -> nullspace projection (:515-546) -> chi2 gate
 *   (UpdaterMSCKF.cpp:739-764) -> stacking (:767-785) -> measurement compression (UpdaterHelper.cpp:548-579)
 *   -> StateHelper::EKFUpdate (StateHelper.cpp:121-202) with R = I.
 * On return P (device) holds the updated covariance; dx_host[n_state] is the correction the caller applies with
 * Type::update; accepted_host[n_feats] / chi2_host[n_feats] mirror the per-feature gate (0 = rejected: the
 * reference sets to_delete and erases it from feature_vec). Any of the host outputs may be NULL. */
int ovp_msckf_update(ovp_ctx *ctx, const ovp_update_opts *opts, double *dx_host, uint8_t *accepted_host,
                     double *chi2_host, ovp_update_info *info);

/* Features the batch format cannot carry - a track of more than OVP_MAX_MEAS observations, observations of a camera other than
 * camera 0 (the reference loops over every camera's measurements of a feature, update/UpdaterHelper.cpp:335-344, and stacks any
 * number of them, update/UpdaterMSCKF.cpp:686-691) - join the SAME update as dense blocks: block k is the feature's system after
 * the nullspace projection, H_k (rows[k] x cols[k], column-major, concatenated in H), its state columns (col_ids, Type::id() +
 * offset, concatenated) and residual (res, concatenated), noise already whitened (R = I).  The call gates every block against the
 * RESIDENT covariance - chi2 = r^T (H P H^T + I)^-1 r <= chi2_multiplier * quantile_0.95(rows), update/UpdaterMSCKF.cpp:739-757 -
 * and keeps the information pair of the accepted ones; the next ovp_msckf_update / ovp_msckf_build_gate_gram_async of the context
 * adds it to the pair of the batch, so batch features and dense blocks are ONE EKF update as in the reference.  Any call that
 * writes the covariance in between drops the pending pair.  Slow path (marginal of the involved columns to the host, gates on
 * the host): meant for the few features per frame that need it.  accepted / chi2: per block, may be NULL. */
int ovp_msckf_dense_blocks(ovp_ctx *ctx, double chi2_multiplier, int n_blocks, const int *rows, const int *cols, const double *H,
                           const int *col_ids, const double *res, uint8_t *accepted, double *chi2);

/* Staged form of the same step, for feature-sharded multi-GPU runs:
 *   stage 1 (per rank, local shard): build + project + gate + local information pair
 *            Ab_dev = [A | b], A = sum_f Hp_f^T Hp_f (n_state x n_state), b = sum_f Hp_f^T r_f,
 *            written to the device buffer returned by ovp_gram_buffer() (f64, (n_state+1) * ld_gram);
 *   (caller all-reduces that buffer over RCCL)
 *   stage 2 (every rank or rank 0): EKF update from the summed pair. */
/* NOTE for callers that touch the pair between the two stages: the update reads Ab only on the columns a point batch involves
 * (the clone poses and the estimated calibration; everything in front of the first of those columns is taken to be zero - the
 * factor of P is then formed in reversed index order and the update runs on the trailing block).  Summing pairs of the same
 * kind over ranks keeps that property; writing other columns does not, and such contributions would be dropped silently -
 * set OVP_POINT_NO_FLIP=1 in the environment (full-width update) for a caller that needs them. */
int ovp_msckf_build_gate_gram_async(ovp_ctx *ctx, const ovp_update_opts *opts);
int ovp_gram_buffer(ovp_ctx *ctx, double **Ab_dev, int *n_rows, int *ld);
int ovp_ekf_update_from_gram_async(ovp_ctx *ctx);
int ovp_msckf_fetch_results(ovp_ctx *ctx, double *dx_host, uint8_t *accepted_host, double *chi2_host,
                            ovp_update_info *info);

/* ---- feature-sharded update over RCCL (no counterpart in the reference, which runs on one core) -----------------
 * One process per GPU, every rank with the same covariance, pose tables and frame resident.  ovp_msckf_update_sharded =
 * ovp_batch_set_range(this rank's balanced share of the features the update is about) -> ovp_msckf_build_gate_gram_async ->
 * ncclAllReduce(sum, f64) of [A | b] on the context's stream -> ovp_ekf_update_from_gram_async -> ovp_msckf_fetch_results: every
 * rank ends with the same covariance and correction (the all-reduced pair is bit-identical on all ranks, the update is
 * deterministic).  nccl_comm is an ncclComm_t of the caller (a C++ host links RCCL itself and owns the communicator); the three
 * helpers below create one for callers that do not (bench.py, the tests): the id of rank 0 travels by whatever means the launcher
 * has.  RCCL is bound with dlopen at first use (OVP_RCCL_LIB overrides the name).  world = 1 / comm = NULL: no collective.
 * accepted / chi2 are filled for this rank's share [*shard_lo, *shard_hi) only (zero elsewhere); with opts->skip_plane_used the
 * share is taken from the features the preceding ovp_msckf_plane_update left. */
typedef struct { char internal[128]; } ovp_rccl_id; /* = ncclUniqueId */
int ovp_rccl_unique_id(ovp_rccl_id *id);
int ovp_rccl_comm_create(const ovp_rccl_id *id, int rank, int world, int device, void **nccl_comm);
int ovp_rccl_comm_destroy(void *nccl_comm);
/* ncclAllReduce(sum, f64) of the pair of ovp_gram_buffer() on the context's stream: the collective between
 * ovp_msckf_build_gate_gram_async and ovp_ekf_update_from_gram_async for callers that drive the stages themselves */
int ovp_rccl_allreduce_gram(ovp_ctx *ctx, void *nccl_comm);
/* the index range [*lo, *hi) of the resident batch that ovp_msckf_update_sharded gives to `rank` of `world` (for callers that drive the
 * staged entries and ovp_batch_set_range themselves): balanced over the features of the update, consecutive ranks tile the batch,
 * lo == hi = an empty share */
int ovp_shard_range(ovp_ctx *ctx, const ovp_update_opts *opts, int rank, int world, int *lo, int *hi);
/* the same arithmetic without a context (pure host code, no device): used[n_feats] = the mask ovp_msckf_plane_update returned
 * (non-zero = consumed by an accepted plane), NULL = no plane loop ran */
int ovp_shard_range_of_mask(const uint8_t *used, int n_feats, int rank, int world, int *lo, int *hi);
/* ERRORS ARE COLLECTIVE: argument / call-order errors are detected before the collective from inputs that are the same on every
 * rank; a rank whose build fails afterwards still enters the all-reduce (zero pair, one more summed f64 word raised), so its peers
 * complete the call and return OVP_E_PEER - no rank is left waiting inside RCCL.  After OVP_E_PEER the covariance of the
 * context is invalid (OVP_E_STATE until the next ovp_cov_upload): the reference treats every failure on this path as fatal
 * (state/StateHelper.cpp:185-187). */
int ovp_msckf_update_sharded(ovp_ctx *ctx, const ovp_update_opts *opts, void *nccl_comm, int rank, int world, double *dx_host,
                             uint8_t *accepted_host, double *chi2_host, ovp_update_info *info, int *shard_lo, int *shard_hi);
/* Completes accepted_host[n_feats] (and chi2_host, may be NULL) of a sharded update on EVERY rank: the shares are disjoint and zero
 * elsewhere, so an ncclAllReduce(sum) is a gather.  For callers that act on every feature's decision on every replica - the
 * Updater surface erases rejected features from feature_vec (update/UpdaterMSCKF.cpp:755-757).  Collective; nccl_comm = NULL: no-op. */
int ovp_rccl_gather_decisions(ovp_ctx *ctx, void *nccl_comm, uint8_t *accepted_host, double *chi2_host);

/* ext ov_core::FeatureInitializerOptions (open_vins ov_core/src/feat/FeatureInitializerOptions.h; not in the reference tree) */
typedef struct {
  int refine_features; /* run single_gaussnewton after the linear triangulation (UpdaterMSCKF.cpp:153-155) */
  int max_runs;
  double init_lamda, max_lamda, min_dx, min_dcost, lam_mult;
  double min_dist, max_dist, max_baseline, max_cond_number;
  int triangulate_1d; /* single_triangulation_1d instead of single_triangulation: depth along the anchor bearing (UpdaterMSCKF.cpp:148-152) */
  int reserved;
} ovp_triang_opts;
void ovp_triang_defaults(ovp_triang_opts *o);

/* ext FeatureInitializer::single_triangulation (+ single_gaussnewton) for every feature of the uploaded batch, against the
 * camera poses of the clones (update/UpdaterMSCKF.cpp:120-166; same block in UpdaterSLAM.cpp:129-144, UpdaterPlane.cpp:139-164).
 * uv_norm [n_feats*max_meas*2] f32 (host) = Feature::uvs_norm in the layout of ovp_feature_batch::uv.  The positions become
 * the linearisation points of the batch on the device (the next update uses them) and are returned in p_FinG_out
 * [n_feats*3] (host, may be NULL); ok[f] = 0 where the reference erases the feature.  Needs ovp_state_upload and a batch. */
int ovp_triangulate(ovp_ctx *ctx, const ovp_triang_opts *opts, const float *uv_norm, double *p_FinG_out, uint8_t *ok);

/* Planes touched by an update (host pointers): plane k (0-based) has reference id k+1.
 * plane_of_feat[f] = 0 for a free point, else the id of the plane feature f lies on (VioManager's feat2plane map,
 * core/VioManager.cpp:516-533). cp / cp_fej: closest-point estimates (State::_features_PLANE value()/fej() for in-state
 * planes, the PlaneFitting estimate otherwise, update/UpdaterMSCKF.cpp:467-475). plane_state_id[k] = Type::id() or -1. */
typedef struct {
  int n_planes;
  const int *plane_of_feat;   /* [n_feats of the uploaded batch] */
  const double *cp;           /* [n_planes*3] */
  const double *cp_fej;       /* [n_planes*3] */
  const int *plane_state_id;  /* [n_planes] */
  /* SLAM landmarks that lie on planes which are NOT in the state (update/UpdaterMSCKF.cpp:232-252), ovp_msckf_plane_update
   * only: each contributes one point-on-plane row whose feature Jacobian stays in the landmark's three state columns
   * (:545-552).  n_slam = 0 / NULL pointers when there are none. */
  int n_slam;                 /* at most OVP_PLANE_MAX_SLAM of them on one plane (OVP_E_CAPACITY beyond) */
  const int *slam_plane;      /* [n_slam] 1-based plane slot */
  const int *slam_state_id;   /* [n_slam] Type::id() of the landmark */
  const double *slam_p;       /* [n_slam*3] Landmark::get_xyz(false) */
  const double *slam_p_fej;   /* [n_slam*3] Landmark::get_xyz(true) */
  /* Diagnostics (ovp_msckf_plane_update only; NULL in production): force_decision[k] = 1 accepts / 0 rejects plane k whatever
   * its chi2 says (the statistic is still computed and returned), any other value lets the gate decide.  The reference has the
   * same switch in spirit (chi2_multipler = 99999 in the simulation configs); with it the parity tests compare states and
   * covariances under the oracle's own accept / reject sequence, separately from the decision statistics. */
  const uint8_t *force_decision; /* [n_planes] */
} ovp_plane_batch;

/* UpdaterMSCKF::update, per-plane loop with MSCKF features (update/UpdaterMSCKF.cpp:411-649): for every plane in
 * ascending id: point-on-plane Jacobians (update/UpdaterHelper.cpp:448-512), UpdaterPlane::nullspace_project_inplace /
 * measurement_compress_inplace (update/UpdaterPlane.cpp:483-552), plane appended (in state) or projected out, plane-level
 * chi2, StateHelper::EKFUpdate.  Planes are sequential: pose tables, calibration, in-state planes and P on the device are
 * updated after every accepted plane.  Needs a batch given with ovp_batch_upload (features with 2..31 observations).
 * Outputs (host, any may be NULL): dx_planes[n_planes*n_state] the correction of every plane (zeros if rejected / skipped)
 * for the caller to apply in order with Type::update; plane_ok; plane_chi2; plane_dof (rows of the reference's test);
 * feat_used[n_feats] = 1 for features consumed by an accepted plane (reference: to_delete + feature_vec_used). */
int ovp_msckf_plane_update(ovp_ctx *ctx, const ovp_update_opts *opts, const ovp_plane_batch *planes, double *dx_planes,
                           uint8_t *plane_ok, double *plane_chi2, int *plane_dof, uint8_t *feat_used);

/* UpdaterPlane::init_vio_plane, core (update/UpdaterPlane.cpp:296-481): for every plane of the batch (none of them in the
 * state; plane_state_id is ignored), in ascending id: stack the on-plane MSCKF features with sigma_c * const_init_multi,
 * project out the feature, compress, StateHelper::initialize(plane, ..., const_init_chi2) = chi2 test of the part that
 * does not involve the plane, covariance augmentation by 3 and EKF update with the remaining rows.
 * Every accepted plane appends 3 columns to the state (new_ids[k] = its Type::id(), else -1); cp_new[3k..] is its
 * initialised value; dx_planes[k*dx_stride ..] the correction of the pre-existing variables to apply with Type::update
 * (dx_stride >= final state size).  Upstream triangulation / plane fitting is the caller's job. */
int ovp_plane_init(ovp_ctx *ctx, const ovp_update_opts *opts, const ovp_plane_batch *planes, double const_init_multi,
                   double const_init_chi2, double *dx_planes, int dx_stride, uint8_t *plane_ok, double *plane_chi2,
                   int *plane_dof, int *new_ids, double *cp_new, uint8_t *feat_used);

/* StateHelper::EKFUpdate (state/StateHelper.cpp:121-202) for a dense H handed over by the host
 * (UpdaterSLAM::update, StateHelper::initialize, merge_planes...): H is [rows x cols] column-major with leading
 * dimension ld, col_ids[cols] gives the state column of every H column, R = I. */
int ovp_ekf_update(ovp_ctx *ctx, const double *H_host, int rows, int cols, int ld, const int *col_ids,
                   const double *res_host, double *dx_host, ovp_update_info *info);

/* Landmarks of the state that were re-observed (UpdaterSLAM::update, update/UpdaterSLAM.cpp:424-673), host pointers.
 * The measurement arrays are laid out as in ovp_feature_batch (one "feature" per landmark).  A landmark held in GLOBAL_3D
 * (feat_rep_slam of every shipped configuration) has its rows built on the device from the tables of ovp_state_upload; one held in
 * an anchored / inverse-depth representation arrives with the dense block [H_x | H_f] the host built (the representation Jacobians
 * of update/UpdaterHelper.cpp:35-193 are host scalar code; for the single inverse depth AFTER the bearing projection of
 * UpdaterSLAM.cpp:499-515) - pre_rows[l] > 0 marks it, the gate and the update are the same. */
typedef struct {
  int n_landmarks;
  int max_meas;               /* row pitch of uv / clone_idx, <= OVP_MAX_MEAS */
  const float *uv;            /* [n_landmarks*max_meas*2] raw pixels of the new observations (Feature::uvs) */
  const int *clone_idx;       /* [n_landmarks*max_meas] clone slot of every observation */
  const int *n_meas;          /* [n_landmarks] */
  const double *p_FinG;       /* [n_landmarks*3] Landmark::get_xyz(false) */
  const double *p_FinG_fej;   /* [n_landmarks*3] Landmark::get_xyz(true) */
  const int *landmark_id;     /* [n_landmarks] Type::id() */
  /* point-on-plane rows (:465-475): Type::id() of the plane of the STATE the landmark lies on, -1 = none; NULL = no planes */
  const int *plane_state_id;  /* [n_landmarks] */
  const double *cp;           /* [n_landmarks*3] State::_features_PLANE.at(..)->value() of that plane */
  const double *cp_fej;       /* [n_landmarks*3] ->fej() */
  /* host-built blocks, NULL when every landmark is GLOBAL_3D */
  const int *pre_rows;        /* [n_landmarks] 0 = rows built on the device */
  const int *pre_cols;        /* [n_landmarks] */
  const double *pre_H;        /* blocks of the marked landmarks in order: [rows x cols] column-major, then res [rows] */
  const int *pre_ids;         /* their state columns, cols per block, in order */
} ovp_slam_batch;

/* UpdaterSLAM::update (update/UpdaterSLAM.cpp:424-673) on the resident covariance: get_feature_jacobian_full for a landmark of the
 * state (UpdaterHelper.cpp:195-513, bearing rows + the point-on-plane rows of :448-512), chi2 test against the marginal covariance
 * (:526-547), the no-plane fallback for a landmark whose plane rows fail (:547-609), stacking (:627-651) and ONE
 * StateHelper::EKFUpdate (:673).  All gates see the covariance in front of the call, as in the reference.
 * status[l]: 0 = rejected (reference: should_marg + to_delete), 1 = accepted, 2 = accepted without its plane
 * (_features_SLAM_to_PLANE[featid] = 0); chi2[l] = statistic of the stage that decided.  dx_host[n_state] for Type::update. */
int ovp_slam_update(ovp_ctx *ctx, const ovp_update_opts *opts, const ovp_slam_batch *batch, double *dx_host, uint8_t *status,
                    double *chi2, ovp_update_info *info);

/* UpdaterSLAM::delayed_init downstream of triangulation (update/UpdaterSLAM.cpp:204-364) for GLOBAL_3D landmarks without plane rows:
 * the candidates of the batch one after the other - get_feature_jacobian_full at the pose tables as the previous candidate left
 * them, StateHelper::initialize (Givens split :434-446, chi2 of the update rows with dof = all rows :464-475,
 * initialize_invertible :489-586, EKFUpdate with the update rows :483-485) - as ONE enqueue with one synchronisation: the loop
 * runs on the device (csrc/k_dinit.hip), three launches per candidate, the device pose tables are updated on the way.
 * Every accepted candidate appends a 3-dof landmark: new_id[l] = its Type::id() after the call (ids are handed out in order over
 * the accepted ones, as the reference does), else -1.  delta_init[3l..] = H_L^-1 res_init, which the caller adds to the new
 * landmark's value (:577); dx[l*dx_stride ..] = the correction of candidate l's EKF update for every variable that was in the
 * state at that point (the landmarks initialised earlier in the call and its own included, at their final ids), to be applied
 * in order with Type::update; zeros for a rejected candidate.  dx_stride >= state size + 3 * n_feats.
 * OVP_E_CAPACITY (nothing touched): the state would outgrow the context, or a track is too long for the one-workgroup S-form -
 * the caller then takes ovp_cov_initialize candidate by candidate. */
int ovp_slam_delayed_init(ovp_ctx *ctx, const ovp_update_opts *opts, const ovp_feature_batch *candidates, uint8_t *ok,
                          double *chi2, int *new_id, double *delta_init, double *dx, int dx_stride);

/* StateHelper::EKFPropagation (state/StateHelper.cpp:41-119): new variables occupy [new_start, new_start+phi_size),
 * Phi is [phi_size x sum(old_sizes)] column-major, Q is [phi_size x phi_size] (upper triangle read). */
int ovp_cov_propagate(ovp_ctx *ctx, int new_start, int phi_size, const int *old_ids, const int *old_sizes, int n_old,
                      const double *Phi_host, const double *Q_host, int *neg_diag);

/* StateHelper::clone (state/StateHelper.cpp:346-396): appends an exact copy of [src_id, src_id+size) at the end, bit for bit
 * as the reference.  The covariance is then positive SEMI-definite until the next propagation; the update entry points fall back
 * to their pivot-dropping / S-form paths when chol(P) meets the zero pivot.
 * ovp_cov_clone_jitter(ctx, rel) > 0 restores the round-2 behaviour (diagonal of the new block stored rel above the copied
 * value, default 0 = off): it keeps such a prior on the fast path at the price of a 1e-11-level departure from the reference. */
int ovp_cov_clone(ovp_ctx *ctx, int src_id, int size);
int ovp_cov_clone_jitter(ovp_ctx *ctx, double relative_inflation);
/* StateHelper::marginalize (state/StateHelper.cpp:276-344): removes rows/cols [id, id+size). */
int ovp_cov_marginalize(ovp_ctx *ctx, int id, int size);
/* StateHelper::augment_clone, time-offset part (state/StateHelper.cpp:613-624):
 *   Cov[:, pose..pose+5] += Cov[:, dt] * dnc_dt^T ;  Cov[pose..pose+5, :] += dnc_dt * Cov[dt, :]   (in that order) */
int ovp_cov_augment_dt(ovp_ctx *ctx, int pose_id, int dt_id, const double dnc_dt[6]);
/* StateHelper::initialize_invertible, covariance part (state/StateHelper.cpp:520-573): appends a k-dimensional variable
 * (k <= 6) at the end of the state.  H_R [k x cols] column-major (ld) with per-column state ids, H_Linv [k x k] and
 * R [k x k] column-major (upper triangle of R read).  New cross-covariance = -P H_R^T H_Linv^T,
 * new block = H_Linv (H_R P H_R^T + R) H_Linv^T.  The caller applies H_Linv * res to the new variable's value. */
int ovp_cov_initialize_invertible(ovp_ctx *ctx, const double *H_R_host, int k, int cols, int ld, const int *col_ids,
                                  const double *H_Linv_host, const double *R_host);
/* StateHelper::initialize downstream of its Givens split (state/StateHelper.cpp:448-487) as ONE device sequence with one
 * synchronisation: Mahalanobis test of the update rows against the prior (:464-475), StateHelper::initialize_invertible with the
 * init rows (:489-586) and StateHelper::EKFUpdate with the update rows (:483-485).  The three separate calls (ovp_cov_marginal,
 * ovp_cov_initialize_invertible, ovp_ekf_update) cost a host round trip each - 0.4 ms per landmark of UpdaterSLAM::delayed_init.
 *   Hx_init [k x cols], H_up [rup x cols] column-major (leading dimensions k and rup), col_ids[cols]; H_Linv, R_init [k x k];
 *   res_up [rup]; noise of the update rows = r_iso * I (every caller whitens to an isotropic R); chi2_threshold =
 *   multiplier * quantile(rows of the FULL residual), as the reference compares.
 * accepted = 0: the test failed, nothing changed.  accepted = 1: the state has k more columns, dx_host[n + k] is the correction of
 * the EKF update (zeros when rup == 0 or do_update == 0); the caller applies H_Linv * res_init to the new variable's value. */
int ovp_cov_initialize(ovp_ctx *ctx, const double *Hx_init, const double *H_up, int k, int rup, int cols, const int *col_ids,
                       const double *H_Linv, const double *R_init, const double *res_up, double r_iso, double chi2_threshold,
                       int do_update, int *accepted, double *chi2, double *dx_host);
/* current covariance dimension */
int ovp_cov_size(ovp_ctx *ctx);

/* 0.95 chi-square quantile used by the gate (boost::math::quantile, update/UpdaterMSCKF.cpp:59-62) */
double ovp_chi2_quantile_095(int dof);

/* ---- plane fitting -------------------------------------------------------------------------------
 * PlaneFitting::plane_fitting (track_plane/PlaneFitting.cpp:84-199) for a batch of planes: RANSAC with the reference's fixed
 * parameters (5-point sets, 200 iterations, 80 % inliers, 5 cm, std::mt19937(8888) + std::shuffle) followed by the refit on the
 * inlier set.  Plane k owns the points [feat_start[k], feat_start[k+1]).  shuffle_variant selects which libstdc++ the
 * hypothesis sets mimic: 0 = GCC <= 10 (the reference's Ubuntu 18.04 / 20.04), 1 = GCC >= 11.
 * Outputs: abcd [n_planes*4] (unit normal, offset), inlier [n_points] (-> feats = best_inliers, :190), ok [n_planes]. */
typedef struct {
  int n_planes;
  const int *feat_start;  /* [n_planes + 1] */
  const double *p_FinG;   /* [n_points*3] Feature::p_FinG */
  int min_inlier_num;     /* StateOptions::plane_msckf_min_feat / plane_init_min_feat */
  double max_cond;        /* StateOptions::plane_msckf_max_cond / plane_init_max_cond */
  int shuffle_variant;
} ovp_planefit_batch;
int ovp_plane_fitting(ovp_ctx *ctx, const ovp_planefit_batch *batch, double *abcd, uint8_t *inlier, uint8_t *ok);

/* PlaneFitting::optimize_plane (track_plane/PlaneFitting.cpp:201-514) for a batch of planes: joint refinement of the plane's
 * closest point and its features over reprojection + point-on-plane factors (ceres/Factor_PointOnPlane.cpp) under
 * CauchyLoss(1), Ceres' dogleg trust-region loop with its default options and at most 12 iterations, then the inlier tests of
 * :456-498.  Observations of feature f are [obs_start[f], obs_start[f] + n_obs[f]); n_obs = 0 marks a SLAM feature (kept
 * constant, :274-279).  R_GtoC / p_CinG are the clonesCAM poses of every observation, uv_norm the normalised measurements.
 * Outputs per plane: cp_out (input value when the call fails), ok, iterations; per feature: p_out (refined position if kept,
 * else the input), kept (-> feats = inliers, :511).  At most 256 features per plane. */
typedef struct {
  int n_planes;
  const int *feat_start;    /* [n_planes + 1] */
  const double *p_FinG;     /* [n_feats*3] */
  const int *obs_start;     /* [n_feats] */
  const int *n_obs;         /* [n_feats] */
  int n_obs_total;
  const double *uv_norm;    /* [n_obs_total*2] */
  const double *R_GtoC;     /* [n_obs_total*9] row-major */
  const double *p_CinG;     /* [n_obs_total*3] */
  const double *cp;         /* [n_planes*3] initial closest points */
  const uint8_t *fix_plane; /* [n_planes] plane is in the state: keep cp constant */
  double sigma_px_norm;     /* sigma_pix / focal length (update/UpdaterMSCKF.cpp:271-272) */
  double sigma_c;           /* StateOptions::sigma_constraint */
  double R_GtoI[9], p_IinG[3]; /* current IMU pose (stateI) */
  double R_ItoC[9], p_IinC[3]; /* camera extrinsics (calib0) */
} ovp_planeopt_batch;
int ovp_plane_optimize(ovp_ctx *ctx, const ovp_planeopt_batch *batch, double *cp_out, double *p_out, uint8_t *kept,
                       uint8_t *ok, int *iterations);

/* ---- diagnostics ---------------------------------------------------------------------------- */
/* copies an internal device buffer to host for tests: name in {"A","b","L","T","Lt","Y","G","rec","chi2",
 * "gramS","syrk"}; returns the byte count copied (or <0). */
long ovp_debug_read(ovp_ctx *ctx, const char *name, void *host, long max_bytes);
/* like ovp_kernel_timer, for the dominant kernel of the plane loop (k_chol2: both factorizations, gate, solve and commit of one
 * plane): enable = 1: HIP events around every k_chol2 launch of ovp_msckf_plane_update AND around the whole loop; enable = 2: around
 * the whole loop only (first launch .. covariance product; read through ovp_host_timing [7]) - events between dependent launches
 * cost microseconds each, so the loop's own time is taken without the per-launch ones */
int ovp_plane_kernel_timer(ovp_ctx *ctx, int enable, int reset, float *avg_ms, int *n_launches);
/* host clock of the two update entry points, accumulated since the last reset (ms): ovp_msckf_plane_update [0] entry -> first
 * launch (grouping of the features by plane, staging tables), [1] entry -> last launch enqueued, [2] wait for the device, [3] calls;
 * ovp_msckf_update [4] enqueue, [5] wait for the published results, [6] calls; [7] the plane loop on the DEVICE clock (HIP events around the whole loop,
 * accumulated only while ovp_plane_kernel_timer is enabled).  bench.py reports them per step. */
int ovp_host_timing(ovp_ctx *ctx, int reset, double *out8);
/* tile Cholesky (the factorization every EKF update and every plane of the plane loop runs) on a host matrix: dense factor of the
 * matrix bordered with brow ((n+1) x (n+1), row-major; brow may be NULL), z = L^-1 brow, y = L^-T z, pivots; avg_ms = average
 * duration over `reps` launches.  n <= ovp_chol2_max_n() (287). */
int ovp_debug_chol2(ovp_ctx *ctx, const double *A_host, int n, int lda, const double *brow_host, int add_identity, double *L_host,
                    double *z_host, double *y_host, double *piv_host, int reps, float *avg_ms);
/* pivot floor of the following ovp_debug_chol2 calls (0 = none): a pivot below it drops its column, as the range part of the
 * plane loop's solve does for the directions a plane's rows do not determine */
void ovp_debug_chol2_floor(double piv_floor);
/* per-stage GPU time of the last update in milliseconds: [0]=build/gate, [1]=gram, [2]=ekf, [3]=total */
int ovp_last_timings(ovp_ctx *ctx, float *ms4);
/* enables hipEvent timing of the dominant kernel; returns avg ms per launch since last reset */
int ovp_kernel_timer(ovp_ctx *ctx, int enable, int reset, float *avg_ms_feat, int *n_launches);

#ifdef __cplusplus
}
#endif
#endif /* OVPLANE_HIP_H */
