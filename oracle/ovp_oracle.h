/*
 * TEST INFRASTRUCTURE ONLY  -  CPU restatement (plain C99, no dependencies) of the rpng/ov_plane
 * MSCKF(+plane) EKF update path, written in the reference's own loop order (dense column-major
 * matrices, sequential Givens rotations, per-feature H*P*H^T + LLT gate, standard P - K*M^T update).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product path (ov_plane_amd/, libovplane_hip.so) never does.
 *
 * PARITY UNPINNED: the reference holds no golden vectors / known-answer tests for this path
 *  and cannot be compiled in this image (every TU needs Eigen3 + Boost + the
 * un-vendored open_vins ov_core @74a63cf).  This restatement is pinned instead by: finite-difference
 * Jacobian checks, algebraic identities, the scipy chi-square table, and agreement with an independent
 * numpy restatement (oracle/np_ref.py).  External semantics (ov_type, quat_ops, CamRadtan, Eigen
 * makeGivens / LLT) follow SURVEY.md Appendix A.
 *
 * All "file:line" citations are relative to /root/reference/ov_plane/src/.
 */
#ifndef OVP_ORACLE_H
#define OVP_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Options read on the path: update/UpdaterOptions.h:37-53, state/StateOptions.h:41-153 */
typedef struct {
  double sigma_px;          /* UpdaterOptions::sigma_pix */
  double chi2_multiplier;   /* UpdaterOptions::chi2_multipler */
  double sigma_constraint;  /* StateOptions::sigma_constraint */
  int do_fej;               /* StateOptions::do_fej */
  int do_calib_camera_pose; /* StateOptions::do_calib_camera_pose */
  int do_calib_camera_intrinsics;
  int reserved;
} ovo_opts;

/* Pose / calibration tables = the ov_type values the path reads (state/State.h:86-121). */
typedef struct {
  int n_state;               /* N = rows of State::_Cov */
  int n_clones;
  const double *clone_q;     /* [n_clones*4] JPL q_GtoI (x y z w) : PoseJPL::Rot()  */
  const double *clone_p;     /* [n_clones*3] p_IinG               : PoseJPL::pos()  */
  const double *clone_q_fej; /* PoseJPL::Rot_fej() */
  const double *clone_p_fej; /* PoseJPL::pos_fej() */
  const int *clone_id;       /* [n_clones] Type::id() (column offset in P) */
  double calib_q[4];         /* R_ItoC  (State::_calib_IMUtoCAM) */
  double calib_p[3];         /* p_IinC */
  int calib_id;              /* -1 when not calibrating */
  double intrinsics[8];      /* fx fy cx cy k1 k2 p1 p2 (State::_cam_intrinsics): radtan, or k1..k4 of the equidistant model */
  int intr_id;
  int cam_fisheye;           /* 0 = ext CamRadtan, 1 = ext CamEqui (State::_cam_intrinsics_cameras) */
} ovo_state;

/* SoA form of UpdaterHelper::UpdaterHelperFeature (update/UpdaterHelper.h:62-105), GLOBAL_3D. */
typedef struct {
  int n_feats;
  int max_meas;
  const float *uv;       /* [n_feats*max_meas*2]  raw pixels, f32 as in the reference */
  const int *clone_idx;  /* [n_feats*max_meas]    index into the clone tables, -1 = padding */
  const int *n_meas;     /* [n_feats] */
  const double *p_FinG;  /* [n_feats*3]  linearisation point (fej == value for MSCKF feats) */
  const double *p_FinG_fej; /* [n_feats*3] or NULL: first-estimate position of in-state (SLAM) landmarks */
} ovo_feats;

/* ---- small building blocks (exported so tests can pin them individually) ---------------------- */
void ovo_quat_2_rot(const double q[4], double R[9]);          /* row-major 3x3; ext quat_ops.h */
void ovo_quat_update(double q[4], const double dth[3]);       /* ext JPLQuat::update */
void ovo_radtan_distort(const double intr[8], const double uvn[2], double uvd[2]);
void ovo_radtan_jacobian(const double intr[8], const double uvn[2], double dz_dzn[4], double dz_dzeta[16]);
void ovo_make_givens(double p, double q, double *c, double *s); /* Eigen JacobiRotation::makeGivens */
double ovo_chi2_quantile_095(int dof);                        /* boost::math::quantile(chi_squared(k),0.95) */
int ovo_llt(double *A, int n, int ld);                        /* in-place lower Cholesky, col-major; 0 ok */

/* update/UpdaterHelper.cpp:195-513.  Column-major outputs, caller-allocated:
 *   H_f [rows x 3 (or 6)], H_x [rows x cols], res [rows]; order_id/order_size [<= n_meas+3].
 * planeid==0: point feature.  planeid!=0: cp/cp_fej used; plane_state_id>=0 means plane in state. */
/* update/UpdaterHelper.cpp:35-193 and its use at :411-421; rep = ext LandmarkRepresentation (0 GLOBAL_3D ... 5
 * ANCHORED_INVERSE_DEPTH_SINGLE), see ovp_oracle.c */
int ovo_feature_jacobian_representation(const ovo_opts *o, const ovo_state *st, int rep, const double p_FinG[3], int anchor_ci,
                                        double *dlam, int *nl_out, double H_anc[18], double H_cal[18]);
int ovo_feature_jacobian_representation_fej(const ovo_opts *o, const ovo_state *st, int rep, const double p_FinG[3],
                                            const double *p_FinG_fej, int anchor_ci, double *dlam, int *nl_out, double H_anc[18],
                                            double H_cal[18]);
int ovo_slam_update_rep(const ovo_opts *o, const ovo_state *st, const ovo_feats *fb, const int *lm_id, const int *plane_of_feat,
                        int n_planes, const double *cp, const double *cp_fej, const int *plane_state_id, double *P, double *dx,
                        uint8_t *accepted, double *chi2, uint8_t *fellback, const int *lm_rep, const int *lm_anchor);
int ovo_feature_jacobian_full_rep(const ovo_opts *o, const ovo_state *st, const ovo_feats *fb, int f, int rep, int anchor_ci,
                                  double *H_f, double *H_x, double *res, int *rows_out, int *cols_out, int *hf_cols_out,
                                  int *order_id, int *order_size, int *n_order_out);
/* update/UpdaterSLAM.cpp:708-850 perform_anchor_change, see ovp_oracle.c */
int ovo_anchor_change(const ovo_opts *o, const ovo_state *st, int rep, int old_ci, int new_ci, int lm_id,
                      const double p_FinA_old[3], const double p_FinA_old_fej[3], double *P, double p_FinA_new[3],
                      double p_FinA_new_fej[3]);
void ovo_equi_distort(const double v[8], const double uvn[2], double uvd[2]);
void ovo_equi_jacobian(const double v[8], const double uvn[2], double dz_dzn[4], double dz_dzeta[16]);
int ovo_feature_jacobian_full(const ovo_opts *o, const ovo_state *st, const ovo_feats *fb, int f, double sigma_c,
                              int planeid, const double *cp, const double *cp_fej, int plane_state_id, double *H_f,
                              double *H_x, double *res, int *rows, int *cols, int *hf_cols, int *order_id,
                              int *order_size, int *n_order);

/* update/UpdaterHelper.cpp:515-546 (H_cp==NULL) and update/UpdaterPlane.cpp:483-517 (H_cp!=NULL).
 * Matrices column-major with leading dimension = rows. On return the first hf_cols rows are the ones to drop
 * (callers read rows [hf_cols, rows)). */
void ovo_nullspace_project(double *H_f, int rows, int hf_cols, double *H_x, int cols, double *H_cp, int cp_cols,
                           double *res);

/* update/UpdaterHelper.cpp:548-579 / update/UpdaterPlane.cpp:519-552. Returns the new row count. */
int ovo_measurement_compress(double *H_x, int rows, int cols, int ld, double *H_cp, int cp_cols, int ld_cp,
                             double *res);

/* state/StateHelper.cpp:231-259 */
void ovo_marginal_cov(const double *P, int n, const int *order_id, const int *order_size, int n_order, double *out);

/* state/StateHelper.cpp:121-202 with dense H [rows x cols] (col-major, ld) in the local order.
 * P updated in place, dx [n] returned. neg_diag set to 1 if any P_ii < 0 (reference would std::exit). */
int ovo_ekf_update(double *P, int n, const int *order_id, const int *order_size, int n_order, const double *H,
                   int rows, int ld, const double *res, double *dx, int *neg_diag);

int ovo_ekf_update_rdiag(double *P, int n, const int *order_id, const int *order_size, int n_order, const double *H,
                         int rows, int ld, const double *res, const double *r_diag /* NULL = I */, double *dx, int *neg_diag);

/* state/StateHelper.cpp:41-119 */
int ovo_ekf_propagation(double *P, int n, int new_start, int phi_size, const int *old_id, const int *old_size,
                        int n_old, const double *Phi, const double *Q, int *neg_diag);

/* update/UpdaterMSCKF.cpp:671-814: the point-feature loop, gate, stack, compress, EKFUpdate.
 * Outputs: dx[n], P in place, accepted[n_feats], chi2[n_feats]. timings[4] (seconds): feat system, compression,
 * update, total.  Returns compressed row count (>=0) or <0 on error. */
int ovo_msckf_point_update(const ovo_opts *o, const ovo_state *st, const ovo_feats *fb, double *P, double *dx,
                           uint8_t *accepted, double *chi2, double *timings);

/* Mutable copy of the filter state the plane loop updates between planes (ext Type::update semantics). */
typedef struct {
  double *clone_q;   /* [n_clones*4] */
  double *clone_p;   /* [n_clones*3] */
  double calib_q[4];
  double calib_p[3];
  double intrinsics[8];
  double *cp;        /* [n_planes*3] plane closest points (only in-state planes are updated) */
} ovo_state_values;

/* ext Type::update for every variable the path touches: clones (PoseJPL), calibration (PoseJPL), intrinsics (Vec),
 * in-state planes (Vec).  FEJ values are never touched.  dx indexed by Type::id(). */
void ovo_apply_dx(const ovo_state *st, int n_planes, const int *plane_state_id, const double *dx, ovo_state_values *val);

/* update/UpdaterMSCKF.cpp:411-649: sequential per-plane updates with MSCKF features
 * (get_feature_jacobian_full with plane rows, UpdaterPlane::nullspace_project_inplace, UpdaterPlane::measurement_compress_inplace,
 *  plane appended (in state) or projected out (not in state), plane-level chi2, StateHelper::EKFUpdate).
 * plane ids are 1..n_planes and processed in ascending order (std::map iteration); plane_of_feat[f] = 0 for free points.
 * cp/cp_fej [n_planes*3]; plane_state_id[k] >= 0 when plane k+1 is in the state.
 * On return: P and val updated, used[f] = 1 for features consumed by an accepted plane, plane_ok[k], plane_chi2[k],
 * plane_rows[k] (rows entering the chi2 test). */
int ovo_msckf_plane_update(const ovo_opts *o, const ovo_state *st, const ovo_feats *fb, const int *plane_of_feat,
                           int n_planes, const double *cp, const double *cp_fej, const int *plane_state_id, double *P,
                           ovo_state_values *val, uint8_t *used, uint8_t *plane_ok, double *plane_chi2, int *plane_rows);
/* All-cores variant of ovo_msckf_point_update (oracle/ovp_oracle_omp.c: OpenMP over the features, TSQR compression).  Context for
 * the speed-up figure only (the reference runs on one thread); n_threads <= 0 = OpenMP's default.  Returns the thread count used
 * (> 0) or < 0 on error. */
int ovo_msckf_point_update_omp(const ovo_opts *o, const ovo_state *st, const ovo_feats *fb, double *P, double *dx,
                               uint8_t *accepted, double *chi2, double *timings, int n_threads);

/* + SLAM landmarks on planes that are not in the state (update/UpdaterMSCKF.cpp:232-252): slam_plane [n_slam] 1-based plane,
 * slam_id [n_slam] Type::id(), slam_p [n_slam*3] values (updated in place), slam_p_fej [n_slam*3] */
int ovo_msckf_plane_update_slam(const ovo_opts *o, const ovo_state *st, const ovo_feats *fb, const int *plane_of_feat,
                                int n_planes, const double *cp, const double *cp_fej, const int *plane_state_id, double *P,
                                ovo_state_values *val, uint8_t *used, uint8_t *plane_ok, double *plane_chi2, int *plane_rows,
                                int n_slam, const int *slam_plane, const int *slam_id, double *slam_p, const double *slam_p_fej);

/* update/UpdaterSLAM.cpp:376-682 (GLOBAL_3D landmarks already in the state, no ArUco): per feature Jacobian with the
 * landmark columns appended, optional point-on-plane rows for in-state planes with the no-plane fallback (:547-609),
 * per-feature chi2, stacking in first-seen order, ONE StateHelper::EKFUpdate without compression.
 * lm_id[f] = Type::id() of the landmark of feature f; plane arrays may be NULL (n_planes = 0).
 * Outputs: dx[n], P in place, accepted[F], chi2[F], fellback[F] (plane constraint dropped for that feature). */
int ovo_slam_update(const ovo_opts *o, const ovo_state *st, const ovo_feats *fb, const int *lm_id, const int *plane_of_feat,
                    int n_planes, const double *cp, const double *cp_fej, const int *plane_state_id, double *P, double *dx,
                    uint8_t *accepted, double *chi2, uint8_t *fellback);

/* update/UpdaterSLAM.cpp:204-364 (core of delayed_init downstream of triangulation): features one by one,
 * StateHelper::initialize(landmark, Hx_order, H_x, H_f, I, res, chi2_multipler); every success appends a 3-dof landmark,
 * updates the state values (val) and P (n grows inside the n_cap storage).  new_id[f] = landmark id or -1,
 * p_out[3f..] = landmark position at the end of the call (initial value + the corrections of later initialisations). */
int ovo_slam_delayed_init(const ovo_opts *o, const ovo_state *st, const ovo_feats *fb, double *P, int n_cap, int *n,
                          ovo_state_values *val, uint8_t *ok, double *chi2, int *new_id, double *p_out);

/* update/UpdaterPlane.cpp:296-481: core of UpdaterPlane::init_vio_plane for planes that already have an estimate
 * (triangulation / RANSAC / Ceres refinement are upstream and out of scope): per plane, stack the on-plane MSCKF features
 * with sigma_c * const_init_multi, nullspace-project the 3 feature columns, compress with H_cp carried, then
 * StateHelper::initialize(plane, ..., const_init_chi2).  Planes are processed in ascending id; every success appends a
 * 3-dof plane to the state (P grows inside its n_cap x n_cap storage) and updates `val`.
 * Outputs per plane: ok, chi2, dof, new_id (Type::id() of the initialised plane or -1), cp_out[3] (its value at the end of
 * the call, i.e. including the corrections it received from planes initialised after it). */
int ovo_plane_init(const ovo_opts *o, const ovo_state *st, const ovo_feats *fb, const int *plane_of_feat, int n_planes,
                   const double *cp_in, double const_init_multi, double const_init_chi2, double *P, int n_cap, int *n,
                   ovo_state_values *val, uint8_t *used, uint8_t *plane_ok, double *plane_chi2, int *plane_dof,
                   int *new_id, double *cp_out);

/* state/StateHelper.cpp:398-487 (initialize) + :489-586 (initialize_invertible), isotropic noise R = r_iso * I.
 * H_R [rows x cols] (col-major, ld = rows) over the variables of `order`, H_L [rows x k] for the new variable (k <= 6),
 * res [rows].  P is [n_cap x n_cap] storage holding the current n x n covariance (leading dimension n_cap); on success
 * it becomes (n+k) x (n+k) and *n is updated.  new_value_update[k] receives H_L^-1 * res_init plus the EKF correction of
 * the new variable; dx[n+k] the EKF correction of the remaining rows (zero when there is no update part).
 * Returns 1 = initialised, 0 = chi2 rejected, <0 error.  chi2/dof of the test are returned when non-NULL. */
int ovo_initialize(double *P, int n_cap, int *n, const int *order_id, const int *order_size, int n_order, double *H_R,
                   double *H_L, int rows, int k, double r_iso, double *res, double chi2_mult, int do_update,
                   double *new_var_delta, double *dx, double *chi2_out, int *dof_out);

/* UpdaterSLAM.cpp:204-364 with StateOptions::feat_rep_slam = rep (0..5), see ovp_oracle.c */
int ovo_slam_delayed_init_rep(const ovo_opts *o, const ovo_state *st, const ovo_feats *fb, int rep, double *P, int n_cap, int *n,
                              ovo_state_values *val, uint8_t *ok, double *chi2_out, int *new_id, double *p_out);

/* ---- state/Propagator.cpp (a11) ------------------------------------------------------------------------------- */
typedef struct {
  double q[4], p[3], v[3], bg[3], ba[3];                 /* State::_imu value (JPL q_GtoI, p_IinG, v_IinG, biases) */
  double q_fej[4], p_fej[3], v_fej[3], bg_fej[3], ba_fej[3];
} ovo_imu_state;

typedef struct {
  double sigma_w, sigma_a, sigma_wb, sigma_ab; /* NoiseManager (continuous-time) */
  double gravity_mag;                          /* _gravity = (0, 0, gravity_mag) */
  int use_rk4;                                 /* StateOptions::use_rk4_integration */
  int imu_avg;                                 /* StateOptions::imu_avg */
  int do_fej;
} ovo_prop_opts;

/* Propagator.cpp:227-341: readings [n x 7] = (t, wm xyz, am xyz) -> the split / interpolated set covering [time0, time1]. */
int ovo_select_imu_readings(const double *imu, int n, double time0, double time1, double *out, int cap);
/* Propagator.cpp:343-454: one IMU interval; advances x, returns F and Qd (col-major 15 x 15, error order th p v bg ba). */
void ovo_predict_and_compute(ovo_imu_state *x, const ovo_prop_opts *po, const double *minus, const double *plus, double *F,
                             double *Qd);
/* Propagator.cpp:37-118: Phi_summed / Qd_summed over the selected readings, propagated mean, last_w (:110-114). */
int ovo_propagate_summed(ovo_imu_state *x, const ovo_prop_opts *po, const double *imu, int n_imu, double time0, double time1,
                         double *Phi, double *Qs, double *last_w, int *n_sel);

/* update/UpdaterZeroVelocity.cpp:68-318, see ovp_oracle.c */
int ovo_zupt_update(const ovo_imu_state *x, const ovo_prop_opts *po, int imu_id, const double *imu, int n_imu, double time0,
                    double time1, double noise_multiplier, double chi2_multiplier, double max_velocity, int disparity_passed,
                    double *P, int n, double *dx, double *chi2_out, int *rows_out);

/* ---- ext ov_core::FeatureInitializer (source not in the reference tree - restated from memory) ------- */
typedef struct {
  int refine_features, max_runs;
  double init_lamda, max_lamda, min_dx, min_dcost, lam_mult, min_dist, max_dist, max_baseline, max_cond_number;
  int triangulate_1d, reserved; /* single_triangulation_1d (update/UpdaterMSCKF.cpp:148-152) */
} ovo_triang_opts;
void ovo_triang_defaults(ovo_triang_opts *o); /* ext FeatureInitializerOptions defaults */
/* single_triangulation (+ single_gaussnewton when refine_features) of every feature of the batch against the camera poses of
 * the clones (update/UpdaterMSCKF.cpp:123-166).  uv_norm [n_feats*max_meas*2] f32 = Feature::uvs_norm.  p_FinG_out
 * [n_feats*3], ok[f] = 0 where the reference drops the feature. */
int ovo_triangulate(const ovo_triang_opts *o, const ovo_state *st, const ovo_feats *fb, const float *uv_norm,
                    double *p_FinG_out, uint8_t *ok);

#ifdef __cplusplus
}
#endif
#endif
