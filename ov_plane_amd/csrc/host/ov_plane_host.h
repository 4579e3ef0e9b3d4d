// Host side of the drop-in: ov_plane's State / StateHelper / UpdaterMSCKF surface (same names, argument meaning and
// error behaviour as the reference) implemented on top of the C-ABI of libovplane_hip.so.  The covariance
// State::_Cov lives on the GPU; everything that is host scalar code in the reference (id bookkeeping, Type::update,
// feature vector side effects) stays here.  Citations relative to /root/reference/ov_plane/src/.
#pragma once
#include <cmath>
#include <array>
#include <map>
#include <mutex>
#include <set>
#include <memory>
#include <unordered_map>
#include <vector>

#include "ov_types.h"
#include "ovplane_hip.h"

namespace ov_plane {

// state/StateOptions.h:41-153 (fields read on this path)
struct StateOptions {
  bool do_fej = true;
  bool do_calib_camera_pose = false;
  bool do_calib_camera_intrinsics = false;
  bool do_calib_camera_timeoffset = false;
  bool imu_avg = false;
  bool use_rk4_integration = true;
  int max_clone_size = 11;
  int num_cameras = 1;
  bool use_plane_constraint = false;
  bool use_plane_constraint_msckf = false;
  bool use_plane_constraint_slamu = false;
  bool use_plane_constraint_slamd = false;
  bool use_plane_slam_feats = false;   // StateOptions.h:105, read by the caller of UpdaterPlane::init_vio_plane (VioManager.cpp:585)
  double sigma_constraint = 0.01;
  double const_init_multi = 1.0;
  double const_init_chi2 = 1.0;
  int max_slam_features = 25;
  int max_aruco_features = 1024;
  double sigma_plane_merge = 0.001;
  double plane_merge_chi2 = 1.00;
  double plane_merge_deg_max = 1.00;
  // StateOptions.h:90-96.  MSCKF features: the projected system is the same for every three-parameter representation (the
  // anchor terms of H_x are H_f_global * dpfg_dx and die with the left nullspace of H_f; tests/test_oracle_pins.py pins this),
  // and ANCHORED_INVERSE_DEPTH_SINGLE is mapped to the MSCKF inverse depth for such features (update/UpdaterMSCKF.cpp:478-481):
  // the device path serves all of them with its GLOBAL_3D arithmetic.  SLAM landmarks live in feat_rep_slam (host-built dense
  // Jacobians, UpdaterSLAM::update / delayed_init / change_anchors).
  ov_type::LandmarkRepresentation::Representation feat_rep_msckf = ov_type::LandmarkRepresentation::GLOBAL_3D;
  ov_type::LandmarkRepresentation::Representation feat_rep_slam = ov_type::LandmarkRepresentation::GLOBAL_3D;
  int max_msckf_plane = 20;            // StateOptions.h:123
  bool use_refine_plane_feat = true;   // StateOptions.h: refine on-plane features and the plane with optimize_plane
  bool use_groundtruths = false;       // debugging aid of the simulation: overwrite fitted planes / their features with the truth (update/UpdaterMSCKF.cpp:284-302,363-380)
  int plane_msckf_min_feat = 20;       // plane_fitting: minimum inliers (MSCKF planes), StateOptions.h:147
  double plane_msckf_max_cond = 100.0; //                 condition number limit of the 5-point solve, :150
  int plane_init_min_feat = 20;        // :141
  double plane_init_max_cond = 100.0;  // :144
  int planefit_shuffle_variant = 0;    // which libstdc++ the RANSAC permutations mimic (0: GCC <= 10, 1: GCC >= 11); not in the reference
  // capacity of the device context (not in the reference: Eigen resizes dynamically)
  int max_state_size = 320;
  int max_features = 8192;
  // HIP device the State's context (covariance, pose tables, feature batches) lives on.  One process per GPU: rank r of a
  // multi-GPU job constructs its replica of the filter with gpu_device = its local rank.  Not in the reference.
  int gpu_device = 0;
};

// update/UpdaterOptions.h:37-53
struct UpdaterOptions {
  double chi2_multipler = 5;
  double sigma_pix = 1;
  double sigma_pix_sq = 1;
};

class StateHelper;

// track_plane/PlaneFitting.h:43-104.  Static like the reference; the device context of the most recently constructed State is
// used (the reference has no such notion: Eigen / Ceres run on the caller's thread).
class PlaneFitting {
public:
  // ext ov_core::FeatureInitializer::ClonePose: camera orientation R_GtoC and position p_CinG
  struct ClonePose {
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    double p[3] = {0, 0, 0};
    const double *Rot() const { return R; }
    const double *pos() const { return p; }
  };
  typedef std::unordered_map<size_t, std::unordered_map<double, ClonePose>> ClonesCam;

  static bool fit_plane(const std::vector<std::shared_ptr<ov_core::Feature>> &feats, double abcd[4], double cond_thresh = 200.0,
                        bool cond_check = true);
  static double point_to_plane_distance(const double point[3], const double abcd[4]) {
    return point[0] * abcd[0] + point[1] * abcd[1] + point[2] * abcd[2] + abcd[3];
  }
  static bool plane_fitting(std::vector<std::shared_ptr<ov_core::Feature>> &feats, double plane_abcd[4], int min_inlier_num,
                            double max_plane_solver_condition_number);
  // stateI = [q_GtoI (JPL), p_IinG], calib0 = [q_ItoC, p_IinC]
  static bool optimize_plane(std::vector<std::shared_ptr<ov_core::Feature>> &feats, double cp_inG[3], ClonesCam &clonesCAM,
                             double sigma_px_norm, double sigma_c, bool fix_plane, const double stateI[7], const double calib0[7]);
  // batched forms the updaters use: one launch for all planes
  static void bind(ovp_ctx *gpu, int shuffle_variant) {
    _gpu = gpu;
    _variant = shuffle_variant;
  }
  static ovp_ctx *bound() { return _gpu; }

private:
  static ovp_ctx *_gpu;
  static int _variant;
};

// state/State.h:48-135
struct StateTestAccess;

class State {
public:
  explicit State(StateOptions &options_);
  ~State();
  double margtimestep() {
    double time = INFINITY;
    for (const auto &clone_imu : _clones_IMU)
      if (clone_imu.first < time) time = clone_imu.first;
    return time;
  }
  int max_covariance_size() { return ovp_cov_size(_gpu); }

  double _timestamp = -1;
  StateOptions _options;
  std::shared_ptr<ov_type::IMU> _imu;
  std::map<double, std::shared_ptr<ov_type::PoseJPL>> _clones_IMU;
  std::shared_ptr<ov_type::Vec> _calib_dt_CAMtoIMU;
  std::unordered_map<size_t, std::shared_ptr<ov_type::PoseJPL>> _calib_IMUtoCAM;
  // lens model of every camera: false = ext CamRadtan, true = ext CamEqui (the reference keeps CamBase objects in
  // _cam_intrinsics_cameras, state/State.h; missing entry = radtan)
  std::unordered_map<size_t, bool> _cam_fisheye;
  std::unordered_map<size_t, std::shared_ptr<ov_type::Vec>> _cam_intrinsics;
  std::unordered_map<size_t, std::shared_ptr<ov_type::Landmark>> _features_SLAM;
  std::unordered_map<size_t, std::shared_ptr<ov_type::Vec>> _features_PLANE;
  std::unordered_map<size_t, size_t> _features_SLAM_to_PLANE;
  // out-of-state plane estimates the caller obtained upstream (PlaneFitting, out of scope here; UpdaterMSCKF.cpp:319-400)
  std::map<size_t, std::vector<double>> _plane_estimates_cp_inG;
  // simulation ground truth (state/State.h:120-121), read by UpdaterMSCKF::update when StateOptions::use_groundtruths is set
  std::unordered_map<size_t, std::array<double, 3>> _true_planes;
  std::unordered_map<size_t, std::array<double, 3>> _true_features;

private:
  friend class StateHelper;
  friend class UpdaterMSCKF;
  friend class UpdaterPlane;
  friend class UpdaterSLAM;
  friend struct StateTestAccess;  // host_capi.cpp harness: swaps a cloned Vec for a Landmark at the same id
  ovp_ctx *_gpu = nullptr;  // replaces Eigen::MatrixXd _Cov (state/State.h:130)
  std::vector<std::shared_ptr<ov_type::Type>> _variables;
};

// state/StateHelper.h:82-243 (subset on the path)
class StateHelper {
public:
  static void EKFPropagation(std::shared_ptr<State> state, const std::vector<std::shared_ptr<ov_type::Type>> &order_NEW,
                             const std::vector<std::shared_ptr<ov_type::Type>> &order_OLD, const MatrixXd &Phi,
                             const MatrixXd &Q);
  static void EKFUpdate(std::shared_ptr<State> state, const std::vector<std::shared_ptr<ov_type::Type>> &H_order,
                        const MatrixXd &H, const VectorXd &res, const MatrixXd &R);
  static void set_initial_covariance(std::shared_ptr<State> state, const MatrixXd &covariance,
                                     const std::vector<std::shared_ptr<ov_type::Type>> &order);
  static MatrixXd get_marginal_covariance(std::shared_ptr<State> state,
                                          const std::vector<std::shared_ptr<ov_type::Type>> &small_variables);
  static MatrixXd get_full_covariance(std::shared_ptr<State> state);
  static void marginalize(std::shared_ptr<State> state, std::shared_ptr<ov_type::Type> marg);
  static std::shared_ptr<ov_type::Type> clone(std::shared_ptr<State> state, std::shared_ptr<ov_type::Type> variable_to_clone);
  static void augment_clone(std::shared_ptr<State> state, const double last_w[3]);
  static void marginalize_old_clone(std::shared_ptr<State> state);
  static void marginalize_slam(std::shared_ptr<State> state);  // state/StateHelper.cpp:638-652
  // state/StateHelper.cpp:654-776: planes whose id was re-assigned by the tracker are merged into their new id (a 3-row
  // EKF update cp_new - cp_old = 0 gated by chi2 and the angle between the normals), planes nobody observes are dropped
  static void merge_planes_and_marginalize(std::shared_ptr<State> state, const std::map<size_t, size_t> &feat2plane,
                                           const std::map<size_t, std::set<size_t>> &plane2oldplane);
  // state/StateHelper.h:172-173 / :188-190
  static bool initialize(std::shared_ptr<State> state, std::shared_ptr<ov_type::Type> new_variable,
                         const std::vector<std::shared_ptr<ov_type::Type>> &H_order, MatrixXd &H_R, MatrixXd &H_L, MatrixXd &R,
                         VectorXd &res, double chi_2_mult, bool do_update = true);
  static void initialize_invertible(std::shared_ptr<State> state, std::shared_ptr<ov_type::Type> new_variable,
                                    const std::vector<std::shared_ptr<ov_type::Type>> &H_order, const MatrixXd &H_R,
                                    const MatrixXd &H_L, const MatrixXd &R, const VectorXd &res);
  // applies a correction to every active variable (the tail of EKFUpdate, state/StateHelper.cpp:190-193)
  static void apply_correction(std::shared_ptr<State> state, const double *dx);

private:
  StateHelper() {}
};

// update/UpdaterHelper.h:55-157: host (dense) versions for the small SLAM / initialisation systems
class UpdaterHelper {
public:
  struct UpdaterHelperFeature {  // update/UpdaterHelper.h:62-105 (the per-camera maps flattened: cam_ids[k] = camera of measurement k)
    size_t featid = 0;
    std::vector<float> uvs;          // [2k]
    std::vector<double> timestamps;  // [k]
    std::vector<int> cam_ids;        // [k], empty = camera 0
    ov_type::LandmarkRepresentation::Representation feat_representation = ov_type::LandmarkRepresentation::GLOBAL_3D;
    int anchor_cam_id = -1;
    double anchor_clone_timestamp = -1;
    double p_FinA[3] = {0, 0, 0}, p_FinA_fej[3] = {0, 0, 0};
    double p_FinG[3] = {0, 0, 0}, p_FinG_fej[3] = {0, 0, 0};
    size_t planeid = 0;
    double cp_FinG[3] = {0, 0, 0}, cp_FinG_fej[3] = {0, 0, 0};
  };
  // update/UpdaterHelper.cpp:35-193: d p_FinG / d (representation parameters) and, for the anchored representations, the
  // blocks w.r.t. the anchor clone (and the extrinsics when they are estimated)
  static void get_feature_jacobian_representation(std::shared_ptr<State> state, UpdaterHelperFeature &feature, MatrixXd &H_f,
                                                  std::vector<MatrixXd> &H_x, std::vector<std::shared_ptr<ov_type::Type>> &x_order);
  // update/UpdaterHelper.cpp:195-513
  static void get_feature_jacobian_full(std::shared_ptr<State> state, UpdaterHelperFeature &feature, double sigma_px, double sigma_c,
                                        MatrixXd &H_f, MatrixXd &H_x, VectorXd &res, std::vector<std::shared_ptr<ov_type::Type>> &x_order);
  // update/UpdaterHelper.cpp:515-546
  static void nullspace_project_inplace(MatrixXd &H_f, MatrixXd &H_x, VectorXd &res);
  // update/UpdaterHelper.cpp:548-579
  static void measurement_compress_inplace(MatrixXd &H_x, VectorXd &res);
};

// update/UpdaterSLAM.h:53-122
class UpdaterSLAM {
public:
  UpdaterSLAM(UpdaterOptions &options_slam, UpdaterOptions &options_aruco, ov_core::FeatureInitializerOptions &feat_init_options);
  // update/UpdaterSLAM.cpp:376-682 (landmarks already in the state)
  void update(std::shared_ptr<State> state, std::vector<std::shared_ptr<ov_core::Feature>> &feature_vec,
              const std::map<size_t, size_t> &feat2plane);
  // update/UpdaterSLAM.cpp:66-374; features with uvs_norm are triangulated on the device first, others carry p_FinG
  void delayed_init(std::shared_ptr<State> state, std::vector<std::shared_ptr<ov_core::Feature>> &feature_vec,
                    const std::map<size_t, size_t> &feat2plane);
  // The same update with every block built on the host (any track length, any number of rows): gates against marginals of the
  // resident covariance (ovp_cov_marginal), one StateHelper::EKFUpdate on the stacked system.  update() takes it when the device
  // entry refuses the batch (a landmark with more than OVP_MAX_MEAS new observations, OVP_E_CAPACITY of the gate kernel): the
  // reference has no size limit on this path.  force_dense_for_tests(true) makes update() take it always.
  void update_dense(std::shared_ptr<State> state, std::vector<std::shared_ptr<ov_core::Feature>> &feature_vec,
                    const std::map<size_t, size_t> &feat2plane);
  static void force_dense_for_tests(bool on) { _force_dense = on; }
  // update/UpdaterSLAM.cpp:684-706: landmarks anchored in the clone that is about to be marginalised move to the newest one
  void change_anchors(std::shared_ptr<State> state);
  // the per-candidate form of delayed_init (host Jacobians, one StateHelper::initialize each): representations other than
  // GLOBAL_3D and candidates with plane rows; delayed_init itself runs GLOBAL_3D candidates as one device loop
  void delayed_init_host_loop(std::shared_ptr<State> state, std::vector<std::shared_ptr<ov_core::Feature>> &feature_vec,
                              const std::map<size_t, size_t> &feat2plane);
  // pose tables of the clone window + camera calibration -> device (ovp_state_upload); clone_slot: timestamp -> clone slot
  static void upload_state_tables(std::shared_ptr<State> state, std::map<double, int> &clone_slot,
                                  std::vector<std::shared_ptr<ov_type::PoseJPL>> &clones);

protected:
  // update/UpdaterSLAM.cpp:708-850: new anchor-frame value of the landmark and covariance propagation with the
  // anchor-change Jacobian  Phi = H_f,new^-1 [H_x,old | H_f,old | -H_x,new]
  void perform_anchor_change(std::shared_ptr<State> state, std::shared_ptr<ov_type::Landmark> landmark, double new_anchor_timestamp,
                             size_t new_cam_id);
  // update/UpdaterSLAM.cpp:120-166: triangulation (+ refinement) of the features that carry normalised measurements
  static void triangulate_on_device(std::shared_ptr<State> state, const ov_core::FeatureInitializerOptions &fio,
                                    std::vector<std::shared_ptr<ov_core::Feature>> &feature_vec);
  UpdaterOptions _options_slam, _options_aruco;
  ov_core::FeatureInitializerOptions _featinit;  // ext FeatureInitializer options (the reference keeps an initializer_feat)
  static bool _force_dense;
  friend struct UpdaterSLAMTestAccess;
};

// update/UpdaterPlane.h:55-123
class UpdaterPlane {
public:
  UpdaterPlane(UpdaterOptions &options, ov_core::FeatureInitializerOptions &feat_init_options);
  // update/UpdaterPlane.cpp:61-481.  Without estimates in state->_plane_estimates_cp_inG the planes that are not in the state
  // are triangulated, fitted and refined here (:76-290) and then initialised from their surviving on-plane MSCKF features;
  // with estimates (pre-fitted entry point) only the initialisation runs.
  void init_vio_plane(std::shared_ptr<State> state, std::vector<std::shared_ptr<ov_core::Feature>> &feature_vec,
                      std::vector<std::shared_ptr<ov_core::Feature>> &feature_vec_used, const std::map<size_t, size_t> &feat2plane);
  // update/UpdaterPlane.cpp:483-517 / :519-552: the UpdaterHelper operations with the plane Jacobian H_cp carried along
  static void nullspace_project_inplace(MatrixXd &H_f, MatrixXd &H_x, MatrixXd &H_cp, VectorXd &res);
  static void measurement_compress_inplace(MatrixXd &H_x, MatrixXd &H_cp, VectorXd &res);

protected:
  UpdaterOptions _options;
  ov_core::FeatureInitializerOptions _featinit;
};

// update/UpdaterMSCKF.h:49-91
class UpdaterMSCKF {
public:
  UpdaterMSCKF(UpdaterOptions &options, ov_core::FeatureInitializerOptions &feat_init_options);
  // Same contract as the reference for everything downstream of triangulation (update/UpdaterMSCKF.cpp:407-828):
  // plane loop first (planes in state / estimates in state->_plane_estimates_cp_inG), then the point loop, chi2 gate,
  // compression and EKF update.  feature_vec: rejected features are erased and every processed feature gets
  // to_delete = true; features consumed by an accepted plane are appended to feature_vec_used.
  // Features must already carry p_FinG (triangulation / plane refinement are upstream and out of scope).
  void update(std::shared_ptr<State> state, std::vector<std::shared_ptr<ov_core::Feature>> &feature_vec,
              std::vector<std::shared_ptr<ov_core::Feature>> &feature_vec_extra,
              std::vector<std::shared_ptr<ov_core::Feature>> &feature_vec_used, const std::map<size_t, size_t> &feat2plane);

  // Multi-GPU (no counterpart in the reference, which runs on one core).  One process per GPU, every process a
  // replica of the filter that is fed the same frames; with a communicator set, update() runs the plane loop on every replica
  // (sequential across planes: update/UpdaterMSCKF.cpp:413-649) and hands the point loop (:695-814) to ovp_msckf_update_sharded:
  // this rank's share of the features -> information pair -> ncclAllReduce -> identical EKF update on every replica; the per-feature
  // gate decisions of the other shares arrive through ovp_rccl_gather_decisions, so feature_vec ends up the same everywhere.
  // nccl_comm: an ncclComm_t the caller owns (or one made by ovp_rccl_comm_create); NULL / world 1 = the single-GPU path.
  void set_communicator(void *nccl_comm, int rank, int world) {
    _comm = nccl_comm;
    _rank = rank;
    _world = world < 1 ? 1 : world;
  }
  // index range of the last point batch this rank built (diagnostics and tests)
  void last_shard(int &lo, int &hi) const {
    lo = _shard_lo;
    hi = _shard_hi;
  }

protected:
  UpdaterOptions _options;
  ov_core::FeatureInitializerOptions _featinit;  // reference: std::shared_ptr<ov_core::FeatureInitializer> initializer_feat
  void *_comm = nullptr;
  int _rank = 0, _world = 1, _shard_lo = 0, _shard_hi = 0;
};

// utils/NoiseManager.h:36-79 (continuous-time IMU noise densities; the *_2 fields are the squares the consumers work with)
struct NoiseManager {
  double sigma_w = 1.6968e-04, sigma_w_2 = 0;
  double sigma_wb = 1.9393e-05, sigma_wb_2 = 0;
  double sigma_a = 2.0000e-3, sigma_a_2 = 0;
  double sigma_ab = 3.0000e-03, sigma_ab_2 = 0;
  // the copy a consumer keeps: every density next to its square (state/Propagator.h:59-64, update/UpdaterZeroVelocity.cpp:49-52
  // both fill the four fields by hand)
  NoiseManager with_squares() const {
    NoiseManager n = *this;
    double *pairs[4][2] = {{&n.sigma_w, &n.sigma_w_2}, {&n.sigma_wb, &n.sigma_wb_2}, {&n.sigma_a, &n.sigma_a_2}, {&n.sigma_ab, &n.sigma_ab_2}};
    for (auto &pr : pairs) *pr[1] = *pr[0] * *pr[0];
    return n;
  }
};

// IMU buffer of Propagator / UpdaterZeroVelocity (state/Propagator.h:70-87, update/UpdaterZeroVelocity.h:83-103): the new reading
// goes to the back; with a horizon (oldest_time != -1) everything more than 0.1 s in front of it is dropped
inline void imu_buffer_push(std::vector<ov_core::ImuData> &buf, const ov_core::ImuData &reading, double oldest_time) {
  buf.push_back(reading);
  if (oldest_time == -1) return;
  const double horizon = oldest_time - 0.10;
  size_t keep = 0;
  for (size_t i = 0; i < buf.size(); ++i)
    if (!(buf[i].timestamp < horizon)) buf[keep++] = buf[i];
  buf.resize(keep);
}

// state/Propagator.h:47-230.  Mean integration and the 15x15 Phi / Qd accumulation are host scalar code as in the
// reference; the covariance step goes to the device through StateHelper::EKFPropagation + augment_clone.
class Propagator {
public:
  Propagator(NoiseManager noises, double gravity_mag);
  void feed_imu(const ov_core::ImuData &message, double oldest_time = -1);                    // Propagator.h:70-87
  void propagate_and_clone(std::shared_ptr<State> state, double timestamp);                  // Propagator.cpp:37-126
  static std::vector<ov_core::ImuData> select_imu_readings(const std::vector<ov_core::ImuData> &imu_data, double time0,
                                                           double time1, bool warn = true);  // Propagator.cpp:227-341
  static ov_core::ImuData interpolate_data(const ov_core::ImuData &imu_1, const ov_core::ImuData &imu_2, double timestamp);
  // F, Qd: column-major 15 x 15 in the IMU error order [th p v bg ba]
  void predict_and_compute(std::shared_ptr<State> state, const ov_core::ImuData &data_minus, const ov_core::ImuData &data_plus,
                           double F[225], double Qd[225]);                                    // Propagator.cpp:343-454
  // last summed transition / noise (diagnostics and tests)
  const double *last_Phi() const { return _Phi; }
  const double *last_Qd() const { return _Qs; }
  const double *last_w() const { return _last_w; }

protected:
  void predict_mean_discrete(std::shared_ptr<State> state, double dt, const double w1[3], const double a1[3], const double w2[3],
                             const double a2[3], double new_q[4], double new_v[3], double new_p[3]);  // :456-488
  void predict_mean_rk4(std::shared_ptr<State> state, double dt, const double w1[3], const double a1[3], const double w2[3],
                        const double a2[3], double new_q[4], double new_v[3], double new_p[3]);       // :490-569
  double last_prop_time_offset = 0.0;
  bool have_last_prop_time_offset = false;
  NoiseManager _noises;
  std::vector<ov_core::ImuData> imu_data;
  std::mutex imu_data_mtx;
  double _gravity[3];
  double _Phi[225], _Qs[225], _last_w[3];
};

// update/UpdaterZeroVelocity.h:59-147.  Detection (chi2 of the raw IMU readings against "standing still", disparity override)
// on the host from the 9x9 marginal of the device covariance; the bias random walk goes through StateHelper::EKFPropagation
// and the 6(n-1)-row update through StateHelper::EKFUpdate, both on the device.
class UpdaterZeroVelocity {
public:
  UpdaterZeroVelocity(UpdaterOptions &options, NoiseManager &noises, std::shared_ptr<ov_core::FeatureDatabase> db,
                      std::shared_ptr<Propagator> prop, double gravity_mag, double zupt_max_velocity, double zupt_noise_multiplier,
                      double zupt_max_disparity);
  void feed_imu(const ov_core::ImuData &message, double oldest_time = -1);  // UpdaterZeroVelocity.h:83-103
  bool try_update(std::shared_ptr<State> state, double timestamp);         // UpdaterZeroVelocity.cpp:68-318
  double last_chi2() const { return _last_chi2; }                          // diagnostics and tests

protected:
  UpdaterOptions _options;
  NoiseManager _noises;
  std::shared_ptr<ov_core::FeatureDatabase> _db;
  std::shared_ptr<Propagator> _prop;
  double _gravity[3];
  double _zupt_max_velocity = 1.0, _zupt_noise_multiplier = 1.0, _zupt_max_disparity = 1.0;
  std::vector<ov_core::ImuData> imu_data;
  double last_prop_time_offset = 0.0;
  bool have_last_prop_time_offset = false;
  double last_zupt_state_timestamp = 0.0;
  double _last_chi2 = 0.0;
};

}  // namespace ov_plane
