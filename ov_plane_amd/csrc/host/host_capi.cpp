// Harness entry point for the host mirror (tests only need a C symbol to drive the C++ classes from ctypes).
// It builds an ov_plane::State the way VioManager would (clone by clone through StateHelper::augment_clone), installs the
// given covariance, wraps the measurements in ov_core::Feature objects and calls UpdaterMSCKF::update.
#include <cstring>

#include "ov_plane_host.h"

using namespace ov_plane;
using namespace ov_type;

// When set, the next ovph_run_msckf_update hands the features over with normalised measurements and NO position, i.e. the
// updater triangulates them itself (update/UpdaterMSCKF.cpp:120-166) before the update.
static const float *g_uv_norm = nullptr;
extern "C" void ovph_set_uv_norm(const float *uv_norm /* [F][M][2] or NULL */) { g_uv_norm = uv_norm; }
// next ovph_run_msckf_update: no plane estimates are handed over, UpdaterMSCKF::update runs plane_fitting / optimize_plane itself
static int g_fit_planes = 0, g_fit_min_feat = 20, g_fit_variant = 0;
static double g_fit_max_cond = 100.0;
// lens model of camera 0 for the next harness call (0 = radtan, 1 = equidistant)
static int g_fisheye = 0;
extern "C" void ovph_set_fisheye(int fisheye) { g_fisheye = fisheye; }
// mode 0 of ovph_run_updater: hold every landmark in this ext LandmarkRepresentation (1..4), anchored in this clone slot
static int g_slam_rep = 0, g_slam_anchor = 0;
extern "C" void ovph_set_slam_rep(int rep, int anchor_ci) {
  g_slam_rep = rep;
  g_slam_anchor = anchor_ci;
}
// mode 1 of ovph_run_updater: StateOptions::feat_rep_slam of the landmarks delayed_init creates
static int g_feat_rep_slam = 0;
extern "C" void ovph_set_feat_rep_slam(int rep) { g_feat_rep_slam = rep; }
// mode 3 of ovph_run_updater: plane of every SLAM landmark (0 = none)
static const int *g_slam_plane = nullptr;
extern "C" void ovph_set_slam_planes(const int *plane_of_landmark) { g_slam_plane = plane_of_landmark; }
// UpdaterSLAM::update takes its dense form (the one a batch the device entry refuses falls back to) until switched off again
extern "C" void ovph_set_slam_force_dense(int on) { UpdaterSLAM::force_dense_for_tests(on != 0); }
extern "C" void ovph_set_plane_fit(int enable, int min_feat, double max_cond, int shuffle_variant) {
  g_fit_planes = enable;
  g_fit_min_feat = min_feat;
  g_fit_max_cond = max_cond;
  g_fit_variant = shuffle_variant;
}

// next ovph_run_msckf_update: a second camera (num_cameras = 2) with these calibration values; cam_of_meas[F][M] says which camera
// took each measurement of the feature batch
static const double *g_cam1_q = nullptr, *g_cam1_p = nullptr, *g_cam1_intr = nullptr;
static const int *g_cam_of_meas = nullptr;
static double g_last_cam1[15];  // [q (4) | p (3) | intrinsics (8)] of camera 1 after the last run that had one
extern "C" void ovph_last_second_camera(double *out15) { memcpy(out15, g_last_cam1, sizeof(g_last_cam1)); }
extern "C" void ovph_set_second_camera(const double *calib_q, const double *calib_p, const double *intr, const int *cam_of_meas) {
  g_cam1_q = calib_q;
  g_cam1_p = calib_p;
  g_cam1_intr = intr;
  g_cam_of_meas = cam_of_meas;
}
// next ovph_run_msckf_update: the State lives on this device and the updater takes the sharded point loop on this communicator
static void *g_comm = nullptr;
static int g_comm_rank = 0, g_comm_world = 1, g_device = 0, g_last_shard[2] = {0, 0};
extern "C" void ovph_set_shard_comm(void *nccl_comm, int rank, int world, int device) {
  g_comm = nccl_comm;
  g_comm_rank = rank;
  g_comm_world = world;
  g_device = device;
}
extern "C" void ovph_last_shard(int *lo, int *hi) {
  *lo = g_last_shard[0];
  *hi = g_last_shard[1];
}

extern "C" int ovph_run_msckf_update(int C, const double *clone_q, const double *clone_p, const double *clone_q_fej,
                                     const double *clone_p_fej, const double *calib_q, const double *calib_p,
                                     const double *intr, int n_planes_in_state, const double *cp_state,
                                     const double *cp_state_fej, const size_t *plane_ids_state, int n_planes_out,
                                     const double *cp_out, const size_t *plane_ids_out, int N, const double *P, int F, int M,
                                     const float *uv, const int *clone_idx, const int *n_meas, const double *p_FinG,
                                     const int *plane_of_feat /* reference plane id or 0 */, double sigma_px,
                                     double chi2_mult, double sigma_c, int do_fej,
                                     /* outputs */ double *out_clone_q, double *out_clone_p, double *out_calib_q,
                                     double *out_calib_p, double *out_intr, double *out_cp_state, double *out_P,
                                     unsigned char *feat_kept, unsigned char *feat_used, unsigned char *feat_deleted) {
  StateOptions so;
  so.do_fej = do_fej != 0;
  so.do_calib_camera_pose = so.do_calib_camera_intrinsics = so.do_calib_camera_timeoffset = true;
  so.max_clone_size = C;
  so.use_plane_constraint = so.use_plane_constraint_msckf = true;
  so.sigma_constraint = sigma_c;
  if (g_fit_planes) {
    so.plane_msckf_min_feat = g_fit_min_feat;
    so.plane_msckf_max_cond = g_fit_max_cond;
    so.planefit_shuffle_variant = g_fit_variant;
  }
  so.max_state_size = N + 8;
  so.max_features = F + 8;
  so.gpu_device = g_device;
  if (g_cam1_q) so.num_cameras = 2;
  auto state = std::make_shared<State>(so);
  state->_cam_fisheye[0] = g_fisheye != 0;
  if (g_cam1_q) {
    VectorXd v(7, 1), iv(8, 1);
    for (int k = 0; k < 4; ++k) v(k) = g_cam1_q[k];
    for (int k = 0; k < 3; ++k) v(4 + k) = g_cam1_p[k];
    for (int k = 0; k < 8; ++k) iv(k) = g_cam1_intr[k];
    state->_calib_IMUtoCAM.at(1)->set_value(v);
    state->_calib_IMUtoCAM.at(1)->set_fej(v);
    state->_cam_intrinsics.at(1)->set_value(iv);
    state->_cam_intrinsics.at(1)->set_fej(iv);
    state->_cam_fisheye[1] = false;
  }
  g_fisheye = 0;
  // calibration values
  {
    VectorXd v(7, 1);
    for (int k = 0; k < 4; ++k) v(k) = calib_q[k];
    for (int k = 0; k < 3; ++k) v(4 + k) = calib_p[k];
    state->_calib_IMUtoCAM.at(0)->set_value(v);
    state->_calib_IMUtoCAM.at(0)->set_fej(v);
    VectorXd iv(8, 1);
    for (int k = 0; k < 8; ++k) iv(k) = intr[k];
    state->_cam_intrinsics.at(0)->set_value(iv);
    state->_cam_intrinsics.at(0)->set_fej(iv);
  }
  // clones, oldest first, through the reference's own entry point
  const double w0[3] = {0, 0, 0};
  std::vector<double> times(C);
  for (int i = 0; i < C; ++i) {
    VectorXd v(7, 1), vf(7, 1);
    for (int k = 0; k < 4; ++k) {
      v(k) = clone_q[4 * i + k];
      vf(k) = clone_q_fej[4 * i + k];
    }
    for (int k = 0; k < 3; ++k) {
      v(4 + k) = clone_p[3 * i + k];
      vf(4 + k) = clone_p_fej[3 * i + k];
    }
    state->_imu->pose()->set_value(v);
    state->_imu->pose()->set_fej(vf);
    times[i] = 100.0 + 0.1 * i;
    state->_timestamp = times[i];
    StateHelper::augment_clone(state, w0);
  }
  // in-state planes: grow the state by cloning the 3-dof position block of the IMU (sizes match), then relabel the new
  // variable's value; the covariance is overwritten below
  std::vector<std::shared_ptr<Vec>> planes;
  for (int k = 0; k < n_planes_in_state; ++k) {
    std::shared_ptr<Type> t = StateHelper::clone(state, state->_imu->p());
    auto pl = std::dynamic_pointer_cast<Vec>(t);
    if (!pl) return -10;
    VectorXd v(3, 1), vf(3, 1);
    for (int a = 0; a < 3; ++a) {
      v(a) = cp_state[3 * k + a];
      vf(a) = cp_state_fej[3 * k + a];
    }
    pl->set_value(v);
    pl->set_fej(vf);
    state->_features_PLANE[plane_ids_state[k]] = pl;
    planes.push_back(pl);
  }
  if (!g_fit_planes)  // otherwise UpdaterMSCKF::update fits / refines the planes itself (UpdaterMSCKF.cpp:196-400)
    for (int k = 0; k < n_planes_out; ++k)
      state->_plane_estimates_cp_inG[plane_ids_out[k]] = {cp_out[3 * k], cp_out[3 * k + 1], cp_out[3 * k + 2]};
  if (state->max_covariance_size() != N) return -11;
  {
    std::vector<std::shared_ptr<Type>> order;
    order.push_back(state->_imu);
    order.push_back(state->_calib_dt_CAMtoIMU);
    order.push_back(state->_calib_IMUtoCAM.at(0));
    order.push_back(state->_cam_intrinsics.at(0));
    if (g_cam1_q) {
      order.push_back(state->_calib_IMUtoCAM.at(1));
      order.push_back(state->_cam_intrinsics.at(1));
    }
    for (auto &c : state->_clones_IMU) order.push_back(c.second);
    for (auto &p : planes) order.push_back(p);
    MatrixXd Pm(N, N);
    memcpy(Pm.data(), P, sizeof(double) * (size_t)N * N);
    StateHelper::set_initial_covariance(state, Pm, order);
  }
  // features
  std::vector<std::shared_ptr<ov_core::Feature>> fv, fextra, fused;
  std::map<size_t, size_t> feat2plane;
  for (int f = 0; f < F; ++f) {
    auto ft = std::make_shared<ov_core::Feature>();
    ft->featid = 5000 + f;
    for (int k = 0; k < n_meas[f]; ++k) {
      ft->timestamps.push_back(times[clone_idx[(size_t)f * M + k]]);
      ft->uvs.push_back(uv[((size_t)f * M + k) * 2]);
      ft->uvs.push_back(uv[((size_t)f * M + k) * 2 + 1]);
      if (g_cam_of_meas) ft->cam_ids.push_back(g_cam_of_meas[(size_t)f * M + k]);
    }
    if (g_uv_norm) {
      for (int k = 0; k < n_meas[f]; ++k) {
        ft->uvs_norm.push_back(g_uv_norm[((size_t)f * M + k) * 2]);
        ft->uvs_norm.push_back(g_uv_norm[((size_t)f * M + k) * 2 + 1]);
      }
    } else {
      memcpy(ft->p_FinG, p_FinG + 3 * f, 3 * sizeof(double));
    }
    if (plane_of_feat[f] > 0) feat2plane[ft->featid] = (size_t)plane_of_feat[f];
    fv.push_back(ft);
  }
  std::vector<std::shared_ptr<ov_core::Feature>> all = fv;
  UpdaterOptions uo;
  uo.sigma_pix = sigma_px;
  uo.chi2_multipler = chi2_mult;
  ov_core::FeatureInitializerOptions fio;
  UpdaterMSCKF updater(uo, fio);
  if (g_comm || g_comm_world > 1) updater.set_communicator(g_comm, g_comm_rank, g_comm_world);
  updater.update(state, fv, fextra, fused, feat2plane);
  updater.last_shard(g_last_shard[0], g_last_shard[1]);
  if (g_cam1_q) {
    memcpy(g_last_cam1, state->_calib_IMUtoCAM.at(1)->quat(), 4 * sizeof(double));
    memcpy(g_last_cam1 + 4, state->_calib_IMUtoCAM.at(1)->pos(), 3 * sizeof(double));
    memcpy(g_last_cam1 + 7, state->_cam_intrinsics.at(1)->value().data(), 8 * sizeof(double));
  }
  g_cam1_q = g_cam1_p = g_cam1_intr = nullptr;
  g_cam_of_meas = nullptr;
  g_comm = nullptr;
  g_comm_rank = 0;
  g_comm_world = 1;
  g_device = 0;
  g_uv_norm = nullptr;
  g_fit_planes = 0;
  // outputs
  int i = 0;
  for (auto &c : state->_clones_IMU) {
    memcpy(out_clone_q + 4 * i, c.second->quat(), 4 * sizeof(double));
    memcpy(out_clone_p + 3 * i, c.second->pos(), 3 * sizeof(double));
    ++i;
  }
  memcpy(out_calib_q, state->_calib_IMUtoCAM.at(0)->quat(), 4 * sizeof(double));
  memcpy(out_calib_p, state->_calib_IMUtoCAM.at(0)->pos(), 3 * sizeof(double));
  memcpy(out_intr, state->_cam_intrinsics.at(0)->value().data(), 8 * sizeof(double));
  for (int k = 0; k < n_planes_in_state; ++k) memcpy(out_cp_state + 3 * k, planes[k]->value().data(), 3 * sizeof(double));
  MatrixXd Pn = StateHelper::get_full_covariance(state);
  memcpy(out_P, Pn.data(), sizeof(double) * (size_t)N * N);
  for (int f = 0; f < F; ++f) {
    feat_kept[f] = 0;
    feat_used[f] = 0;
    feat_deleted[f] = all[f]->to_delete ? 1 : 0;
  }
  for (auto &ft : fv) feat_kept[ft->featid - 5000] = 1;
  for (auto &ft : fused) feat_used[ft->featid - 5000] = 1;
  return 0;
}


// Harness for StateHelper::initialize: state with C clones + calibration, covariance P (N x N), a new Vec(k) variable.
// order_ids[n_order] select the measuring variables by Type::id() (clones / calibration / intrinsics).
extern "C" int ovph_run_initialize(int C, const double *clone_q, const double *clone_p, int N, const double *P, int n_order,
                                   const int *order_ids, int rows, int cols, const double *H_R, int k, const double *H_L,
                                   const double *res, double r_iso, double chi2_mult, const double *new_value0,
                                   double *out_P /* (N+k)^2 */, double *out_new_value, double *out_clone_q,
                                   double *out_clone_p, double *out_calib_p, double *out_intr) {
  StateOptions so;
  so.do_calib_camera_pose = so.do_calib_camera_intrinsics = so.do_calib_camera_timeoffset = true;
  so.max_clone_size = C;
  so.max_state_size = N + 16;
  so.max_features = 16;
  auto state = std::make_shared<State>(so);
  const double w0[3] = {0, 0, 0};
  for (int i = 0; i < C; ++i) {
    VectorXd v(7, 1);
    for (int q = 0; q < 4; ++q) v(q) = clone_q[4 * i + q];
    for (int q = 0; q < 3; ++q) v(4 + q) = clone_p[3 * i + q];
    state->_imu->pose()->set_value(v);
    state->_imu->pose()->set_fej(v);
    state->_timestamp = 100.0 + 0.1 * i;
    StateHelper::augment_clone(state, w0);
  }
  if (state->max_covariance_size() != N) return -11;
  std::vector<std::shared_ptr<Type>> all;
  all.push_back(state->_imu);
  all.push_back(state->_calib_dt_CAMtoIMU);
  all.push_back(state->_calib_IMUtoCAM.at(0));
  all.push_back(state->_cam_intrinsics.at(0));
  for (auto &c : state->_clones_IMU) all.push_back(c.second);
  {
    MatrixXd Pm(N, N);
    memcpy(Pm.data(), P, sizeof(double) * (size_t)N * N);
    StateHelper::set_initial_covariance(state, Pm, all);
  }
  std::vector<std::shared_ptr<Type>> H_order;
  for (int i = 0; i < n_order; ++i) {
    std::shared_ptr<Type> found;
    for (auto &v : all)
      if (v->id() == order_ids[i]) found = v;
    if (!found) return -12;
    H_order.push_back(found);
  }
  MatrixXd HR(rows, cols), HL(rows, k), R = MatrixXd::Zero(rows, rows);
  VectorXd r(rows, 1);
  memcpy(HR.data(), H_R, sizeof(double) * (size_t)rows * cols);
  memcpy(HL.data(), H_L, sizeof(double) * (size_t)rows * k);
  memcpy(r.data(), res, sizeof(double) * rows);
  for (int i = 0; i < rows; ++i) R(i, i) = r_iso;
  auto nv = std::make_shared<Vec>(k);
  VectorXd v0(k, 1);
  for (int i = 0; i < k; ++i) v0(i) = new_value0[i];
  nv->set_value(v0);
  nv->set_fej(v0);
  const bool ok = StateHelper::initialize(state, nv, H_order, HR, HL, R, r, chi2_mult, true);
  const int n2 = state->max_covariance_size();
  MatrixXd Pn = StateHelper::get_full_covariance(state);
  memcpy(out_P, Pn.data(), sizeof(double) * (size_t)n2 * n2);
  memcpy(out_new_value, nv->value().data(), sizeof(double) * k);
  int i = 0;
  for (auto &c : state->_clones_IMU) {
    memcpy(out_clone_q + 4 * i, c.second->quat(), 4 * sizeof(double));
    memcpy(out_clone_p + 3 * i, c.second->pos(), 3 * sizeof(double));
    ++i;
  }
  memcpy(out_calib_p, state->_calib_IMUtoCAM.at(0)->pos(), 3 * sizeof(double));
  memcpy(out_intr, state->_cam_intrinsics.at(0)->value().data(), 8 * sizeof(double));
  return ok ? 1 : 0;
}

// ---- shared state construction for the SLAM / plane-init harnesses ----------------------------------
namespace ov_plane {
struct StateTestAccess {
  static void replace_last_variable(std::shared_ptr<State> state, std::shared_ptr<Type> v) { state->_variables.back() = v; }
};
}  // namespace ov_plane

namespace {
struct HarnessState {
  std::shared_ptr<State> state;
  std::vector<double> times;
  std::vector<std::shared_ptr<Landmark>> landmarks;
  std::vector<std::shared_ptr<Vec>> planes;
};

// Order of the state: imu, dt, calib, intrinsics, clones (oldest first), SLAM landmarks, planes (= synth.state_layout).
int build_harness_state(HarnessState &hs, StateOptions &so, int C, const double *clone_q, const double *clone_p,
                        const double *clone_q_fej, const double *clone_p_fej, const double *calib_q, const double *calib_p,
                        const double *intr, int n_slam, const double *slam_p, const double *slam_p_fej, int n_planes,
                        const double *cp, const double *cp_fej, int N, const double *P) {
  hs.state = std::make_shared<State>(so);
  auto &state = hs.state;
  state->_cam_fisheye[0] = g_fisheye != 0;
  g_fisheye = 0;
  VectorXd v(7, 1);
  for (int k = 0; k < 4; ++k) v(k) = calib_q[k];
  for (int k = 0; k < 3; ++k) v(4 + k) = calib_p[k];
  state->_calib_IMUtoCAM.at(0)->set_value(v);
  state->_calib_IMUtoCAM.at(0)->set_fej(v);
  VectorXd iv(8, 1);
  for (int k = 0; k < 8; ++k) iv(k) = intr[k];
  state->_cam_intrinsics.at(0)->set_value(iv);
  state->_cam_intrinsics.at(0)->set_fej(iv);
  const double w0[3] = {0, 0, 0};
  hs.times.resize(C);
  for (int i = 0; i < C; ++i) {
    VectorXd a(7, 1), af(7, 1);
    for (int k = 0; k < 4; ++k) {
      a(k) = clone_q[4 * i + k];
      af(k) = clone_q_fej[4 * i + k];
    }
    for (int k = 0; k < 3; ++k) {
      a(4 + k) = clone_p[3 * i + k];
      af(4 + k) = clone_p_fej[3 * i + k];
    }
    state->_imu->pose()->set_value(a);
    state->_imu->pose()->set_fej(af);
    hs.times[i] = 100.0 + 0.1 * i;
    state->_timestamp = hs.times[i];
    StateHelper::augment_clone(state, w0);
  }
  // 3-dof variables: grow the covariance through StateHelper::clone (the block is overwritten below), then swap the cloned
  // Vec for the variable type the updaters expect at the same id
  for (int k = 0; k < n_slam; ++k) {
    std::shared_ptr<Type> t = StateHelper::clone(state, state->_imu->p());
    auto lm = std::make_shared<Landmark>(3);
    lm->set_local_id(t->id());
    lm->_featid = 9000 + k;
    lm->set_from_xyz(slam_p + 3 * k, false);
    lm->set_from_xyz(slam_p_fej + 3 * k, true);
    StateTestAccess::replace_last_variable(state, lm);
    state->_features_SLAM[lm->_featid] = lm;
    hs.landmarks.push_back(lm);
  }
  for (int k = 0; k < n_planes; ++k) {
    auto pl = std::dynamic_pointer_cast<Vec>(StateHelper::clone(state, state->_imu->p()));
    if (!pl) return -10;
    VectorXd a(3, 1), af(3, 1);
    for (int q = 0; q < 3; ++q) {
      a(q) = cp[3 * k + q];
      af(q) = cp_fej[3 * k + q];
    }
    pl->set_value(a);
    pl->set_fej(af);
    state->_features_PLANE[(size_t)(k + 1)] = pl;
    hs.planes.push_back(pl);
  }
  if (state->max_covariance_size() != N) return -11;
  std::vector<std::shared_ptr<Type>> order;
  order.push_back(state->_imu);
  order.push_back(state->_calib_dt_CAMtoIMU);
  order.push_back(state->_calib_IMUtoCAM.at(0));
  order.push_back(state->_cam_intrinsics.at(0));
  for (auto &c : state->_clones_IMU) order.push_back(c.second);
  for (auto &l : hs.landmarks) order.push_back(l);
  for (auto &p : hs.planes) order.push_back(p);
  MatrixXd Pm(N, N);
  memcpy(Pm.data(), P, sizeof(double) * (size_t)N * N);
  StateHelper::set_initial_covariance(state, Pm, order);
  return 0;
}

std::vector<std::shared_ptr<ov_core::Feature>> make_features(const HarnessState &hs, int F, int M, const float *uv, const int *clone_idx,
                                                             const int *n_meas, const double *p_FinG, size_t id0) {
  std::vector<std::shared_ptr<ov_core::Feature>> fv;
  for (int f = 0; f < F; ++f) {
    auto ft = std::make_shared<ov_core::Feature>();
    ft->featid = id0 + f;
    for (int k = 0; k < n_meas[f]; ++k) {
      ft->timestamps.push_back(hs.times[clone_idx[(size_t)f * M + k]]);
      ft->uvs.push_back(uv[((size_t)f * M + k) * 2]);
      ft->uvs.push_back(uv[((size_t)f * M + k) * 2 + 1]);
    }
    if (g_uv_norm) {  // features as the tracker hands them over: normalised measurements, no position yet
      for (int k = 0; k < n_meas[f]; ++k) {
        ft->uvs_norm.push_back(g_uv_norm[((size_t)f * M + k) * 2]);
        ft->uvs_norm.push_back(g_uv_norm[((size_t)f * M + k) * 2 + 1]);
      }
    } else if (p_FinG) {
      memcpy(ft->p_FinG, p_FinG + 3 * f, 3 * sizeof(double));
    }
    fv.push_back(ft);
  }
  return fv;
}

void export_state(const HarnessState &hs, double *out_clone_q, double *out_clone_p, double *out_calib_q, double *out_calib_p,
                  double *out_intr, double *out_slam_p, double *out_cp, double *out_P, int *out_n) {
  auto &state = hs.state;
  int i = 0;
  for (auto &c : state->_clones_IMU) {
    memcpy(out_clone_q + 4 * i, c.second->quat(), 4 * sizeof(double));
    memcpy(out_clone_p + 3 * i, c.second->pos(), 3 * sizeof(double));
    ++i;
  }
  memcpy(out_calib_q, state->_calib_IMUtoCAM.at(0)->quat(), 4 * sizeof(double));
  memcpy(out_calib_p, state->_calib_IMUtoCAM.at(0)->pos(), 3 * sizeof(double));
  memcpy(out_intr, state->_cam_intrinsics.at(0)->value().data(), 8 * sizeof(double));
  for (size_t k = 0; k < hs.landmarks.size(); ++k) memcpy(out_slam_p + 3 * k, hs.landmarks[k]->value().data(), 3 * sizeof(double));
  for (size_t k = 0; k < hs.planes.size(); ++k) memcpy(out_cp + 3 * k, hs.planes[k]->value().data(), 3 * sizeof(double));
  const int n2 = state->max_covariance_size();
  MatrixXd Pn = StateHelper::get_full_covariance(state);
  memcpy(out_P, Pn.data(), sizeof(double) * (size_t)n2 * n2);
  *out_n = n2;
}
}  // namespace

// mode 3: UpdaterMSCKF::update on a state that also holds SLAM landmarks (ids 9000 + k, planes from ovph_set_slam_planes) and
// n_planes in-state planes (ids 1..n_planes); the features carry normalised measurements, the updater triangulates, fits and
// refines (ovph_set_plane_fit must be on).
// mode 0: UpdaterSLAM::update (feature f observes landmark f, n_slam == F); mode 1: UpdaterSLAM::delayed_init (n_slam == 0,
// out_new_p [F*3] / out_new_id [F] describe the landmarks that joined the state); mode 2: UpdaterPlane::init_vio_plane
// (n_planes in-state planes must be 0; cp_out_est [n_planes_out*3] are the upstream plane estimates with ids 1..n).
extern "C" int ovph_run_updater(int mode, int C, const double *clone_q, const double *clone_p, const double *clone_q_fej,
                                const double *clone_p_fej, const double *calib_q, const double *calib_p, const double *intr,
                                int n_slam, const double *slam_p, const double *slam_p_fej, int n_planes, const double *cp,
                                const double *cp_fej, int n_planes_out, const double *cp_out_est, int N, const double *P, int F,
                                int M, const float *uv, const int *clone_idx, const int *n_meas, const double *p_FinG,
                                const int *plane_of_feat, double sigma_px, double chi2_mult, double sigma_c, int do_fej,
                                double const_init_multi, double const_init_chi2, int n_cap,
                                /* outputs */ double *out_clone_q, double *out_clone_p, double *out_calib_q, double *out_calib_p,
                                double *out_intr, double *out_slam_p, double *out_cp, double *out_P, int *out_n,
                                unsigned char *feat_kept, unsigned char *feat_deleted, unsigned char *lm_should_marg,
                                int *slam_to_plane, double *out_new_p, int *out_new_id) {
  StateOptions so;
  so.do_fej = do_fej != 0;
  so.do_calib_camera_pose = so.do_calib_camera_intrinsics = so.do_calib_camera_timeoffset = true;
  so.max_clone_size = C;
  so.use_plane_constraint = so.use_plane_constraint_msckf = so.use_plane_constraint_slamu = so.use_plane_constraint_slamd = true;
  so.sigma_constraint = sigma_c;
  so.const_init_multi = const_init_multi;
  so.const_init_chi2 = const_init_chi2;
  if (g_fit_planes) {
    so.plane_init_min_feat = so.plane_msckf_min_feat = g_fit_min_feat;
    so.plane_init_max_cond = so.plane_msckf_max_cond = g_fit_max_cond;
    so.planefit_shuffle_variant = g_fit_variant;
  }
  so.max_state_size = n_cap;
  so.max_features = F + 8;
  HarnessState hs;
  int rc = build_harness_state(hs, so, C, clone_q, clone_p, clone_q_fej, clone_p_fej, calib_q, calib_p, intr, n_slam, slam_p,
                               slam_p_fej, n_planes, cp, cp_fej, N, P);
  if (rc) return rc;
  auto &state = hs.state;
  const size_t id0 = (mode == 0) ? 9000 : 5000;
  auto fv = make_features(hs, F, M, uv, clone_idx, n_meas, p_FinG, id0);
  auto all = fv;
  std::map<size_t, size_t> feat2plane;
  for (int f = 0; f < F; ++f)
    if (plane_of_feat && plane_of_feat[f] > 0) feat2plane[id0 + f] = (size_t)plane_of_feat[f];
  UpdaterOptions uo, ua;
  uo.sigma_pix = sigma_px;
  uo.chi2_multipler = chi2_mult;
  ua = uo;
  ov_core::FeatureInitializerOptions fio;
  if (mode == 3) {
    for (int k = 0; k < n_slam && g_slam_plane; ++k)
      if (g_slam_plane[k] > 0) feat2plane[9000 + k] = (size_t)g_slam_plane[k];
    UpdaterMSCKF up(uo, fio);
    std::vector<std::shared_ptr<ov_core::Feature>> fextra, fused;
    up.update(state, fv, fextra, fused, feat2plane);
    g_fit_planes = 0;
    g_uv_norm = nullptr;
    g_slam_plane = nullptr;
    std::set<size_t> left;
    for (auto &ft : fv) left.insert(ft->featid);
    export_state(hs, out_clone_q, out_clone_p, out_calib_q, out_calib_p, out_intr, out_slam_p, out_cp, out_P, out_n);
    for (int f = 0; f < F; ++f) {
      feat_kept[f] = left.count(id0 + f) ? 1 : 0;
      feat_deleted[f] = all[f]->to_delete ? 1 : 0;
      slam_to_plane[f] = -1;
    }
    for (int k = 0; k < n_slam && k < F; ++k) {
      auto it = state->_features_SLAM_to_PLANE.find(9000 + k);
      if (it != state->_features_SLAM_to_PLANE.end()) slam_to_plane[k] = (int)it->second;
    }
    return 0;
  }
  if (mode == 0 && g_slam_rep != 0) {
    // re-express the landmarks (built as GLOBAL_3D from their global value / first estimate) in the requested representation
    auto calib = state->_calib_IMUtoCAM.at(0);
    auto anchor = state->_clones_IMU.at(hs.times[g_slam_anchor]);
    for (auto &lm : hs.landmarks) {
      double pg[3], pgf[3], pa[3], paf[3], t[3];
      lm->get_xyz(false, pg);
      lm->get_xyz(true, pgf);
      lm->_feat_representation = (LandmarkRepresentation::Representation)g_slam_rep;
      if (LandmarkRepresentation::is_relative_representation(lm->_feat_representation)) {
        lm->_anchor_cam_id = 0;
        lm->_anchor_clone_timestamp = hs.times[g_slam_anchor];
        for (int pass = 0; pass < 2; ++pass) {
          const double *R = pass ? anchor->Rot_fej() : anchor->Rot(), *pp = pass ? anchor->pos_fej() : anchor->pos();
          const double *src = pass ? pgf : pg;
          double *dst = pass ? paf : pa;
          const double d[3] = {src[0] - pp[0], src[1] - pp[1], src[2] - pp[2]};
          for (int i = 0; i < 3; ++i) t[i] = R[3 * i] * d[0] + R[3 * i + 1] * d[1] + R[3 * i + 2] * d[2];
          for (int i = 0; i < 3; ++i)
            dst[i] = calib->Rot()[3 * i] * t[0] + calib->Rot()[3 * i + 1] * t[1] + calib->Rot()[3 * i + 2] * t[2] + calib->pos()[i];
        }
        lm->set_from_xyz(pa, false);
        lm->set_from_xyz(paf, true);
      } else {
        lm->set_from_xyz(pg, false);
        lm->set_from_xyz(pgf, true);
      }
    }
    g_slam_rep = 0;
  }
  if (mode == 0) {
    UpdaterSLAM up(uo, ua, fio);
    up.update(state, fv, feat2plane);
  } else if (mode == 1) {
    UpdaterSLAM up(uo, ua, fio);
    state->_options.feat_rep_slam = (LandmarkRepresentation::Representation)g_feat_rep_slam;
    g_feat_rep_slam = 0;
    up.delayed_init(state, fv, feat2plane);
    for (int f = 0; f < F; ++f) {
      out_new_id[f] = -1;
      auto it = state->_features_SLAM.find(id0 + f);
      if (it == state->_features_SLAM.end()) continue;
      out_new_id[f] = it->second->id();
      memcpy(out_new_p + 3 * f, it->second->value().data(), (size_t)it->second->size() * sizeof(double));
    }
  } else {
    if (!g_fit_planes)  // otherwise init_vio_plane triangulates, fits and refines itself (UpdaterPlane.cpp:76-290)
      for (int k = 0; k < n_planes_out; ++k)
        state->_plane_estimates_cp_inG[(size_t)(k + 1)] = {cp_out_est[3 * k], cp_out_est[3 * k + 1], cp_out_est[3 * k + 2]};
    UpdaterPlane up(uo, fio);
    std::vector<std::shared_ptr<ov_core::Feature>> used;
    up.init_vio_plane(state, fv, used, feat2plane);
    g_fit_planes = 0;
    g_uv_norm = nullptr;
    for (int k = 0; k < n_planes_out; ++k) {
      out_new_id[k] = -1;
      auto it = state->_features_PLANE.find((size_t)(k + 1));
      if (it == state->_features_PLANE.end()) continue;
      out_new_id[k] = it->second->id();
      memcpy(out_new_p + 3 * k, it->second->value().data(), 3 * sizeof(double));
    }
  }
  export_state(hs, out_clone_q, out_clone_p, out_calib_q, out_calib_p, out_intr, out_slam_p, out_cp, out_P, out_n);
  for (int f = 0; f < F; ++f) {
    feat_kept[f] = 0;
    feat_deleted[f] = all[f]->to_delete ? 1 : 0;
    slam_to_plane[f] = -1;
    auto it = state->_features_SLAM_to_PLANE.find(id0 + f);
    if (it != state->_features_SLAM_to_PLANE.end()) slam_to_plane[f] = (int)it->second;
  }
  for (auto &ft : fv) feat_kept[ft->featid - id0] = 1;
  for (size_t k = 0; k < hs.landmarks.size(); ++k) lm_should_marg[k] = hs.landmarks[k]->should_marg ? 1 : 0;
  return 0;
}

// Propagator::propagate_and_clone harness: state with C clones (+ calibration), IMU state x16 / x16_fej, covariance P,
// n_imu readings rows (t, wm, am).  state->_timestamp = t_state, camera time offset dt_cam_imu; propagates to `timestamp`.
extern "C" int ovph_run_propagate(int C, const double *clone_q, const double *clone_p, const double *imu_x16, const double *imu_x16_fej,
                                  double calib_dt, int N, const double *P, int n_imu, const double *imu, double t_state,
                                  double timestamp, const double *sigmas4 /* w a wb ab */, double gravity_mag, int use_rk4,
                                  int imu_avg, int do_fej,
                                  /* outputs */ double *out_x16, double *out_x16_fej, double *out_Phi, double *out_Qd, double *out_last_w,
                                  double *out_P /* (N+6)^2 */, double *out_new_clone7) {
  StateOptions so;
  so.do_fej = do_fej != 0;
  so.do_calib_camera_pose = so.do_calib_camera_intrinsics = so.do_calib_camera_timeoffset = true;
  so.max_clone_size = C + 1;
  so.use_rk4_integration = use_rk4 != 0;
  so.imu_avg = imu_avg != 0;
  so.max_state_size = N + 16;
  so.max_features = 16;
  auto state = std::make_shared<State>(so);
  const double w0[3] = {0, 0, 0};
  for (int i = 0; i < C; ++i) {
    VectorXd v(7, 1);
    for (int q = 0; q < 4; ++q) v(q) = clone_q[4 * i + q];
    for (int q = 0; q < 3; ++q) v(4 + q) = clone_p[3 * i + q];
    state->_imu->pose()->set_value(v);
    state->_imu->pose()->set_fej(v);
    state->_timestamp = t_state - 0.1 * (C - i);
    StateHelper::augment_clone(state, w0);
  }
  if (state->max_covariance_size() != N) return -11;
  VectorXd x(16, 1), xf(16, 1), dtv(1, 1);
  for (int k = 0; k < 16; ++k) {
    x(k) = imu_x16[k];
    xf(k) = imu_x16_fej[k];
  }
  state->_imu->set_value(x);
  state->_imu->set_fej(xf);
  dtv(0) = calib_dt;
  state->_calib_dt_CAMtoIMU->set_value(dtv);
  state->_calib_dt_CAMtoIMU->set_fej(dtv);
  state->_timestamp = t_state;
  std::vector<std::shared_ptr<Type>> all;
  all.push_back(state->_imu);
  all.push_back(state->_calib_dt_CAMtoIMU);
  all.push_back(state->_calib_IMUtoCAM.at(0));
  all.push_back(state->_cam_intrinsics.at(0));
  for (auto &c : state->_clones_IMU) all.push_back(c.second);
  MatrixXd Pm(N, N);
  memcpy(Pm.data(), P, sizeof(double) * (size_t)N * N);
  StateHelper::set_initial_covariance(state, Pm, all);
  NoiseManager nm;
  nm.sigma_w = sigmas4[0];
  nm.sigma_a = sigmas4[1];
  nm.sigma_wb = sigmas4[2];
  nm.sigma_ab = sigmas4[3];
  Propagator prop(nm, gravity_mag);
  for (int i = 0; i < n_imu; ++i) {
    ov_core::ImuData d;
    d.timestamp = imu[7 * i];
    for (int k = 0; k < 3; ++k) {
      d.wm[k] = imu[7 * i + 1 + k];
      d.am[k] = imu[7 * i + 4 + k];
    }
    prop.feed_imu(d);
  }
  prop.propagate_and_clone(state, timestamp);
  memcpy(out_x16, state->_imu->value().data(), 16 * sizeof(double));
  memcpy(out_x16_fej, state->_imu->fej().data(), 16 * sizeof(double));
  memcpy(out_Phi, prop.last_Phi(), 225 * sizeof(double));
  memcpy(out_Qd, prop.last_Qd(), 225 * sizeof(double));
  memcpy(out_last_w, prop.last_w(), 3 * sizeof(double));
  const int n2 = state->max_covariance_size();
  if (n2 != N + 6) return -12;
  MatrixXd Pn = StateHelper::get_full_covariance(state);
  memcpy(out_P, Pn.data(), sizeof(double) * (size_t)n2 * n2);
  auto nc = state->_clones_IMU.at(timestamp);
  memcpy(out_new_clone7, nc->value().data(), 7 * sizeof(double));
  return 0;
}

// UpdaterZeroVelocity::try_update harness: the state of ovph_run_propagate, n_calls (1 or 2) consecutive camera times
// timestamps[k]; the feature database holds n_tracks tracks seen at t_state (uv0), timestamps[0] (uv1) and, with two calls,
// timestamps[1] (uv1 again).  Outputs per call: accepted, chi2; final IMU value, calib_dt, covariance, state timestamp, and
// how many measurements the database still holds at timestamps[0] (cleanup_measurements_exact on the second acceptance).
extern "C" int ovph_run_zupt(int C, const double *clone_q, const double *clone_p, const double *imu_x16, const double *imu_x16_fej,
                             double calib_dt, int N, const double *P, int n_imu, const double *imu, double t_state, int n_calls,
                             const double *timestamps, const double *sigmas4, double gravity_mag, int do_fej, double noise_mult,
                             double chi2_mult, double max_velocity, double max_disparity, int n_tracks, const float *uv0,
                             const float *uv1,
                             /* outputs */ int *out_accepted, double *out_chi2, double *out_x16, double *out_calib_dt, double *out_P,
                             double *out_timestamp, int *out_meas_at_t1) {
  StateOptions so;
  so.do_fej = do_fej != 0;
  so.do_calib_camera_pose = so.do_calib_camera_intrinsics = so.do_calib_camera_timeoffset = true;
  so.max_clone_size = C + 1;
  so.max_state_size = N + 16;
  so.max_features = 16;
  auto state = std::make_shared<State>(so);
  const double w0[3] = {0, 0, 0};
  for (int i = 0; i < C; ++i) {
    VectorXd v(7, 1);
    for (int q = 0; q < 4; ++q) v(q) = clone_q[4 * i + q];
    for (int q = 0; q < 3; ++q) v(4 + q) = clone_p[3 * i + q];
    state->_imu->pose()->set_value(v);
    state->_imu->pose()->set_fej(v);
    state->_timestamp = t_state - 0.1 * (C - i);
    StateHelper::augment_clone(state, w0);
  }
  if (state->max_covariance_size() != N) return -11;
  VectorXd x(16, 1), xf(16, 1), dtv(1, 1);
  for (int k = 0; k < 16; ++k) {
    x(k) = imu_x16[k];
    xf(k) = imu_x16_fej[k];
  }
  state->_imu->set_value(x);
  state->_imu->set_fej(xf);
  dtv(0) = calib_dt;
  state->_calib_dt_CAMtoIMU->set_value(dtv);
  state->_calib_dt_CAMtoIMU->set_fej(dtv);
  state->_timestamp = t_state;
  std::vector<std::shared_ptr<Type>> all;
  all.push_back(state->_imu);
  all.push_back(state->_calib_dt_CAMtoIMU);
  all.push_back(state->_calib_IMUtoCAM.at(0));
  all.push_back(state->_cam_intrinsics.at(0));
  for (auto &c : state->_clones_IMU) all.push_back(c.second);
  MatrixXd Pm(N, N);
  memcpy(Pm.data(), P, sizeof(double) * (size_t)N * N);
  StateHelper::set_initial_covariance(state, Pm, all);
  NoiseManager nm;
  nm.sigma_w = sigmas4[0];
  nm.sigma_a = sigmas4[1];
  nm.sigma_wb = sigmas4[2];
  nm.sigma_ab = sigmas4[3];
  auto prop = std::make_shared<Propagator>(nm, gravity_mag);
  auto db = std::make_shared<ov_core::FeatureDatabase>();
  for (int f = 0; f < n_tracks; ++f) {
    db->update_feature(100 + f, t_state, 0, uv0[2 * f], uv0[2 * f + 1], 0.f, 0.f);
    for (int k = 0; k < n_calls; ++k) db->update_feature(100 + f, timestamps[k], 0, uv1[2 * f], uv1[2 * f + 1], 0.f, 0.f);
  }
  UpdaterOptions uo;
  uo.chi2_multipler = chi2_mult;
  UpdaterZeroVelocity zupt(uo, nm, db, prop, gravity_mag, max_velocity, noise_mult, max_disparity);
  for (int i = 0; i < n_imu; ++i) {
    ov_core::ImuData d;
    d.timestamp = imu[7 * i];
    for (int k = 0; k < 3; ++k) {
      d.wm[k] = imu[7 * i + 1 + k];
      d.am[k] = imu[7 * i + 4 + k];
    }
    zupt.feed_imu(d);
  }
  for (int k = 0; k < n_calls; ++k) {
    out_accepted[k] = zupt.try_update(state, timestamps[k]) ? 1 : 0;
    out_chi2[k] = zupt.last_chi2();
  }
  memcpy(out_x16, state->_imu->value().data(), 16 * sizeof(double));
  *out_calib_dt = state->_calib_dt_CAMtoIMU->value()(0);
  if (state->max_covariance_size() != N) return -12;
  MatrixXd Pn = StateHelper::get_full_covariance(state);
  memcpy(out_P, Pn.data(), sizeof(double) * (size_t)N * N);
  *out_timestamp = state->_timestamp;
  *out_meas_at_t1 = (int)db->features_containing(timestamps[0]).size();
  return 0;
}

// StateHelper::marginalize_slam + merge_planes_and_marginalize harness.  State: clones, n_slam landmarks (should_marg flags),
// n_planes in-state planes (ids 1..n).  merge_pairs [n_pairs x 2] = (old id, new id); active_planes = ids observed by features.
// Outputs: final P / n, for every plane id 1..n+8 its Type::id() (or -1) and value, landmark ids (or -1).
extern "C" int ovph_run_state_maintenance(int C, const double *clone_q, const double *clone_p, const double *calib_q,
                                          const double *calib_p, const double *intr, int n_slam, const double *slam_p,
                                          const unsigned char *should_marg, int n_planes, const double *cp, int N, const double *P,
                                          int n_pairs, const int *merge_pairs, int n_active, const int *active_planes,
                                          double sigma_plane_merge, double plane_merge_chi2, double plane_merge_deg_max,
                                          double *out_P, int *out_n, int *out_plane_state_id /* [n_planes+8] */,
                                          double *out_plane_cp /* [(n_planes+8)*3] */, int *out_slam_state_id,
                                          int *out_slam_to_plane) {
  StateOptions so;
  so.do_calib_camera_pose = so.do_calib_camera_intrinsics = so.do_calib_camera_timeoffset = true;
  so.max_clone_size = C;
  so.use_plane_constraint = true;
  so.sigma_plane_merge = sigma_plane_merge;
  so.plane_merge_chi2 = plane_merge_chi2;
  so.plane_merge_deg_max = plane_merge_deg_max;
  so.max_state_size = N + 8;
  so.max_features = 16;
  HarnessState hs;
  int rc = build_harness_state(hs, so, C, clone_q, clone_p, clone_q, clone_p, calib_q, calib_p, intr, n_slam, slam_p, slam_p,
                               n_planes, cp, cp, N, P);
  if (rc) return rc;
  auto &state = hs.state;
  for (int k = 0; k < n_slam; ++k) {
    hs.landmarks[k]->should_marg = should_marg[k] != 0;
    state->_features_SLAM_to_PLANE[hs.landmarks[k]->_featid] = 1;
  }
  StateHelper::marginalize_slam(state);
  std::map<size_t, size_t> feat2plane;
  for (int k = 0; k < n_active; ++k) feat2plane[7000 + k] = (size_t)active_planes[k];
  std::map<size_t, std::set<size_t>> plane2oldplane;
  for (int k = 0; k < n_pairs; ++k) plane2oldplane[(size_t)merge_pairs[2 * k + 1]].insert((size_t)merge_pairs[2 * k]);
  StateHelper::merge_planes_and_marginalize(state, feat2plane, plane2oldplane);
  const int n2 = state->max_covariance_size();
  MatrixXd Pn = StateHelper::get_full_covariance(state);
  memcpy(out_P, Pn.data(), sizeof(double) * (size_t)n2 * n2);
  *out_n = n2;
  for (int k = 0; k < n_planes + 8; ++k) {
    out_plane_state_id[k] = -1;
    auto it = state->_features_PLANE.find((size_t)(k + 1));
    if (it == state->_features_PLANE.end()) continue;
    out_plane_state_id[k] = it->second->id();
    memcpy(out_plane_cp + 3 * k, it->second->value().data(), 3 * sizeof(double));
  }
  for (int k = 0; k < n_slam; ++k) {
    auto it = state->_features_SLAM.find(hs.landmarks[k]->_featid);
    out_slam_state_id[k] = (it == state->_features_SLAM.end()) ? -1 : it->second->id();
    out_slam_to_plane[k] = state->_features_SLAM_to_PLANE.count(hs.landmarks[k]->_featid) ? 1 : 0;
  }
  return 0;
}

// UpdaterPlane::nullspace_project_inplace / measurement_compress_inplace and the UpdaterHelper twins on dense inputs
// (column-major).  Returns the number of rows left; outputs overwrite the leading rows of the inputs (ld unchanged).
extern "C" int ovph_run_plane_givens(int op /* 0 nullspace, 1 compress */, int rows, int hf_cols, double *H_f, int cols,
                                     double *H_x, int cp_cols, double *H_cp, double *res) {
  MatrixXd Hf(rows, std::max(hf_cols, 1)), Hx(rows, cols), Hcp(rows, std::max(cp_cols, 1));
  VectorXd r(rows, 1);
  if (hf_cols) memcpy(Hf.data(), H_f, sizeof(double) * (size_t)rows * hf_cols);
  memcpy(Hx.data(), H_x, sizeof(double) * (size_t)rows * cols);
  if (cp_cols) memcpy(Hcp.data(), H_cp, sizeof(double) * (size_t)rows * cp_cols);
  memcpy(r.data(), res, sizeof(double) * rows);
  if (op == 0) {
    if (cp_cols) UpdaterPlane::nullspace_project_inplace(Hf, Hx, Hcp, r);
    else UpdaterHelper::nullspace_project_inplace(Hf, Hx, r);
  } else {
    if (cp_cols) UpdaterPlane::measurement_compress_inplace(Hx, Hcp, r);
    else UpdaterHelper::measurement_compress_inplace(Hx, r);
  }
  const int ro = Hx.rows();
  for (int j = 0; j < cols; ++j)
    for (int i = 0; i < ro; ++i) H_x[(size_t)j * rows + i] = Hx(i, j);
  for (int j = 0; j < cp_cols; ++j)
    for (int i = 0; i < ro; ++i) H_cp[(size_t)j * rows + i] = Hcp(i, j);
  for (int i = 0; i < ro; ++i) res[i] = r(i);
  return ro;
}

// ovph_run_sequence extras: per-frame IMU value [K][16] and IMU pose covariance [K][36] (column-major) after each frame; with
// uv_norm [F][M][2] the features arrive untriangulated (uvs_norm set, no position) and UpdaterMSCKF::update triangulates them
static double *g_seq_traj = nullptr, *g_seq_posecov = nullptr;
static const float *g_seq_uv_norm = nullptr;
extern "C" void ovph_set_sequence_trace(double *traj, double *posecov, const float *uv_norm) {
  g_seq_traj = traj;
  g_seq_posecov = posecov;
  g_seq_uv_norm = uv_norm;
}
// ... planar regularities in the loop: plane id (0 = none) of every feature of the concatenated list -> feat2plane.
// mode 1: point-on-plane constraints with planes estimated per update (UpdaterMSCKF, planes stay out of the state);
// mode 2: additionally UpdaterPlane::init_vio_plane before the update (core/VioManager.cpp:583-588), planes join the state
// (the final covariance is then larger than N and not returned; out_planes_in_state [1] receives how many planes it holds)
static const int *g_seq_plane = nullptr;
static int g_seq_plane_mode = 0, g_seq_plane_min_feat = 20;
static int *g_seq_planes_in_state = nullptr;
static double g_seq_sigma_c = 0.01;
extern "C" void ovph_set_sequence_planes(const int *plane_of_feat, int mode, int min_feat, double sigma_c, int *out_planes_in_state) {
  g_seq_plane = plane_of_feat;
  g_seq_plane_mode = mode;
  g_seq_plane_min_feat = min_feat;
  g_seq_sigma_c = sigma_c;
  g_seq_planes_in_state = out_planes_in_state;
}

// Closed loop over several camera frames with the reference's own call order (core/VioManager.cpp:348 propagate_and_clone,
// :670 UpdaterMSCKF::update, :864-866 marginalize_old_clone): state with C clones, IMU state, covariance P (N = 30 + 6 C).
// Frame k: propagate + clone to frame_time[k], update with that frame's features (slots index the C+1 clones of the window,
// oldest first; positions given), marginalize the oldest clone.  Outputs the final filter state and covariance.
extern "C" int ovph_run_sequence(int C, const double *clone_q, const double *clone_p, const double *clone_q_fej,
                                 const double *clone_p_fej, const double *calib_q, const double *calib_p, const double *intr,
                                 const double *imu_x16, const double *imu_x16_fej, double calib_dt, int N, const double *P,
                                 int n_imu, const double *imu, double t_state, const double *sigmas4, double gravity_mag,
                                 int use_rk4, int do_fej, int K, const double *frame_time, const int *feat_offset /* [K+1] */,
                                 int M, const float *uv, const int *clone_slot, const int *n_meas, const double *p_FinG,
                                 double sigma_px, double chi2_mult,
                                 /* outputs */ double *out_clone_q, double *out_clone_p, double *out_x16, double *out_calib_q,
                                 double *out_calib_p, double *out_intr, double *out_dt, double *out_P, int *n_kept_per_frame) {
  StateOptions so;
  so.do_fej = do_fej != 0;
  so.do_calib_camera_pose = so.do_calib_camera_intrinsics = so.do_calib_camera_timeoffset = true;
  so.max_clone_size = C;
  so.use_rk4_integration = use_rk4 != 0;
  so.max_state_size = N + 16 + 3 * 16;
  so.max_features = 4096;
  const int plane_mode = g_seq_plane ? g_seq_plane_mode : 0;
  if (plane_mode) {
    so.use_plane_constraint = so.use_plane_constraint_msckf = true;
    so.use_plane_constraint_slamu = so.use_plane_constraint_slamd = true;
    so.use_plane_slam_feats = plane_mode == 2;
    so.sigma_constraint = g_seq_sigma_c;
    so.plane_init_min_feat = so.plane_msckf_min_feat = g_seq_plane_min_feat;
  }
  auto state = std::make_shared<State>(so);
  {
    VectorXd v(7, 1);
    for (int k = 0; k < 4; ++k) v(k) = calib_q[k];
    for (int k = 0; k < 3; ++k) v(4 + k) = calib_p[k];
    state->_calib_IMUtoCAM.at(0)->set_value(v);
    state->_calib_IMUtoCAM.at(0)->set_fej(v);
    VectorXd iv(8, 1);
    for (int k = 0; k < 8; ++k) iv(k) = intr[k];
    state->_cam_intrinsics.at(0)->set_value(iv);
    state->_cam_intrinsics.at(0)->set_fej(iv);
  }
  const double w0[3] = {0, 0, 0};
  for (int i = 0; i < C; ++i) {
    VectorXd a(7, 1), af(7, 1);
    for (int k = 0; k < 4; ++k) {
      a(k) = clone_q[4 * i + k];
      af(k) = clone_q_fej[4 * i + k];
    }
    for (int k = 0; k < 3; ++k) {
      a(4 + k) = clone_p[3 * i + k];
      af(4 + k) = clone_p_fej[3 * i + k];
    }
    state->_imu->pose()->set_value(a);
    state->_imu->pose()->set_fej(af);
    state->_timestamp = t_state - 0.1 * (C - i);
    StateHelper::augment_clone(state, w0);
  }
  if (state->max_covariance_size() != N) return -11;
  VectorXd x(16, 1), xf(16, 1), dtv(1, 1);
  for (int k = 0; k < 16; ++k) {
    x(k) = imu_x16[k];
    xf(k) = imu_x16_fej[k];
  }
  state->_imu->set_value(x);
  state->_imu->set_fej(xf);
  dtv(0) = calib_dt;
  state->_calib_dt_CAMtoIMU->set_value(dtv);
  state->_calib_dt_CAMtoIMU->set_fej(dtv);
  state->_timestamp = t_state;
  {
    std::vector<std::shared_ptr<Type>> all;
    all.push_back(state->_imu);
    all.push_back(state->_calib_dt_CAMtoIMU);
    all.push_back(state->_calib_IMUtoCAM.at(0));
    all.push_back(state->_cam_intrinsics.at(0));
    for (auto &c : state->_clones_IMU) all.push_back(c.second);
    MatrixXd Pm(N, N);
    memcpy(Pm.data(), P, sizeof(double) * (size_t)N * N);
    StateHelper::set_initial_covariance(state, Pm, all);
  }
  NoiseManager nm;
  nm.sigma_w = sigmas4[0];
  nm.sigma_a = sigmas4[1];
  nm.sigma_wb = sigmas4[2];
  nm.sigma_ab = sigmas4[3];
  Propagator prop(nm, gravity_mag);
  for (int i = 0; i < n_imu; ++i) {
    ov_core::ImuData d;
    d.timestamp = imu[7 * i];
    for (int k = 0; k < 3; ++k) {
      d.wm[k] = imu[7 * i + 1 + k];
      d.am[k] = imu[7 * i + 4 + k];
    }
    prop.feed_imu(d);
  }
  UpdaterOptions uo;
  uo.sigma_pix = sigma_px;
  uo.chi2_multipler = chi2_mult;
  ov_core::FeatureInitializerOptions fio;
  UpdaterMSCKF updater(uo, fio);
  UpdaterPlane updater_plane(uo, fio);
  std::map<size_t, size_t> feat2plane;
  for (int k = 0; k < K; ++k) {
    feat2plane.clear();
    prop.propagate_and_clone(state, frame_time[k]);  // VioManager.cpp:348
    std::vector<double> times;
    for (auto &c : state->_clones_IMU) times.push_back(c.first);
    std::vector<std::shared_ptr<ov_core::Feature>> fv, fextra, fused;
    for (int f = feat_offset[k]; f < feat_offset[k + 1]; ++f) {
      auto ft = std::make_shared<ov_core::Feature>();
      ft->featid = 1000 + f;
      for (int q = 0; q < n_meas[f]; ++q) {
        ft->timestamps.push_back(times.at(clone_slot[(size_t)f * M + q]));
        ft->uvs.push_back(uv[((size_t)f * M + q) * 2]);
        ft->uvs.push_back(uv[((size_t)f * M + q) * 2 + 1]);
      }
      if (g_seq_uv_norm) {
        for (int q = 0; q < 2 * n_meas[f]; ++q) ft->uvs_norm.push_back(g_seq_uv_norm[(size_t)f * M * 2 + q]);
      } else {
        memcpy(ft->p_FinG, p_FinG + 3 * f, 3 * sizeof(double));
      }
      fv.push_back(ft);
      if (plane_mode && g_seq_plane[f] > 0) feat2plane[ft->featid] = (size_t)g_seq_plane[f];
    }
    if (plane_mode == 2) {  // VioManager.cpp:543-600: planar candidates first try to initialise their planes
      std::vector<std::shared_ptr<ov_core::Feature>> fplane, finit_used;
      for (auto &ft : fv)
        if (feat2plane.count(ft->featid)) fplane.push_back(ft);
      updater_plane.init_vio_plane(state, fplane, finit_used, feat2plane);
      std::set<size_t> used;
      for (auto &ft : finit_used) used.insert(ft->featid);
      std::vector<std::shared_ptr<ov_core::Feature>> rest;
      for (auto &ft : fv)
        if (!used.count(ft->featid)) rest.push_back(ft);
      fv.swap(rest);
    }
    updater.update(state, fv, fextra, fused, feat2plane);  // VioManager.cpp:670
    n_kept_per_frame[k] = (int)fv.size();
    StateHelper::marginalize_old_clone(state);  // VioManager.cpp:864-866
    if (g_seq_traj) memcpy(g_seq_traj + 16 * (size_t)k, state->_imu->value().data(), 16 * sizeof(double));
    if (g_seq_posecov) {
      std::vector<std::shared_ptr<Type>> po;
      po.push_back(state->_imu->pose());
      MatrixXd Pp = StateHelper::get_marginal_covariance(state, po);
      memcpy(g_seq_posecov + 36 * (size_t)k, Pp.data(), 36 * sizeof(double));
    }
  }
  g_seq_traj = g_seq_posecov = nullptr;
  g_seq_uv_norm = nullptr;
  if (g_seq_planes_in_state) *g_seq_planes_in_state = (int)state->_features_PLANE.size();
  g_seq_plane = nullptr;
  g_seq_planes_in_state = nullptr;
  int i = 0;
  for (auto &c : state->_clones_IMU) {
    memcpy(out_clone_q + 4 * i, c.second->quat(), 4 * sizeof(double));
    memcpy(out_clone_p + 3 * i, c.second->pos(), 3 * sizeof(double));
    ++i;
  }
  if (i != C) return -13;
  memcpy(out_x16, state->_imu->value().data(), 16 * sizeof(double));
  memcpy(out_calib_q, state->_calib_IMUtoCAM.at(0)->quat(), 4 * sizeof(double));
  memcpy(out_calib_p, state->_calib_IMUtoCAM.at(0)->pos(), 3 * sizeof(double));
  memcpy(out_intr, state->_cam_intrinsics.at(0)->value().data(), 8 * sizeof(double));
  *out_dt = state->_calib_dt_CAMtoIMU->value()(0);
  if (plane_mode == 2) return 0;  // planes joined the state: the covariance is larger than the caller's buffer
  if (state->max_covariance_size() != N) return -12;
  MatrixXd Pn = StateHelper::get_full_covariance(state);
  memcpy(out_P, Pn.data(), sizeof(double) * (size_t)N * N);
  return 0;
}


// ---- on-disk formats (ov_plane_io.h) ---------------------------------------------------------------------------------
#include "ov_plane_io.h"

#include <fstream>
#include <sstream>

// no device needed: timing CSV (header + one row) into buf; returns the length
extern "C" int ovph_format_timing(int use_plane, int max_slam, const double *vals9, char *buf, int cap) {
  ov_plane::StateOptions so;
  so.use_plane_constraint = use_plane != 0;
  so.max_slam_features = max_slam;
  ov_plane::TimingRecord r;
  r.timestamp_inI = vals9[0];
  r.track = vals9[1];
  r.prop = vals9[2];
  r.planeinit = vals9[3];
  r.msckf = vals9[4];
  r.slam_update = vals9[5];
  r.slam_delay = vals9[6];
  r.marg = vals9[7];
  r.total = vals9[8];
  std::ostringstream os;
  ov_plane::write_timing_header(os, so);
  ov_plane::write_timing_row(os, so, r);
  const std::string s = os.str();
  if ((int)s.size() + 1 > cap) return -1;
  memcpy(buf, s.c_str(), s.size() + 1);
  return (int)s.size();
}

// no device needed: returns the number of poses (<= cap) copied to out [cap][8], or -1
extern "C" int ovph_load_trajectory(const char *path, double *out, int cap) {
  std::vector<std::array<double, 8>> poses;
  if (!ov_plane::load_trajectory(path, poses)) return -1;
  const int n = std::min((int)poses.size(), cap);
  for (int i = 0; i < n; ++i) memcpy(out + 8 * i, poses[i].data(), 8 * sizeof(double));
  return (int)poses.size();
}

// per-frame trace of every following UpdaterMSCKF::update (point update) into `path`; NULL / "" stops tracing
extern "C" int ovph_update_trace(const char *path) { return ov_plane::open_update_trace(path ? path : "") ? 0 : -1; }

// no device needed: reads every frame of `in` with the C++ reader and writes it back to `out` with the C++ writer
extern "C" int ovph_trace_copy(const char *in, const char *out) {
  std::ifstream is(in, std::ios::binary);
  std::ofstream os(out, std::ios::binary);
  if (!is.is_open() || !os.is_open()) return -1;
  int n = 0;
  ov_plane::FrameTrace f;
  while (is.peek() != EOF) {
    if (!ov_plane::read_frame_trace(is, f, n == 0)) return -2;
    if (!ov_plane::write_frame_trace(os, f, n == 0)) return -3;
    ++n;
  }
  return n;
}

// needs the device: a state as ovph_run_msckf_update builds it (no planes), one line of each of the three files
extern "C" int ovph_format_state_files(int C, const double *clone_q, const double *clone_p, const double *calib_q, const double *calib_p,
                                       const double *intr, int N, const double *P, double timestamp, double dt, int with_gt,
                                       char *est, char *sd, char *gt, int cap) {
  using namespace ov_plane;
  StateOptions so;
  so.do_calib_camera_pose = so.do_calib_camera_intrinsics = so.do_calib_camera_timeoffset = true;
  so.max_clone_size = C;
  so.max_state_size = N + 8;
  so.max_features = 16;
  auto state = std::make_shared<State>(so);
  {
    VectorXd v(7, 1);
    for (int k = 0; k < 4; ++k) v(k) = calib_q[k];
    for (int k = 0; k < 3; ++k) v(4 + k) = calib_p[k];
    state->_calib_IMUtoCAM.at(0)->set_value(v);
    VectorXd iv(8, 1);
    for (int k = 0; k < 8; ++k) iv(k) = intr[k];
    state->_cam_intrinsics.at(0)->set_value(iv);
    VectorXd d(1, 1);
    d(0) = dt;
    state->_calib_dt_CAMtoIMU->set_value(d);
  }
  const double w0[3] = {0, 0, 0};
  for (int i = 0; i < C; ++i) {
    VectorXd v(7, 1);
    for (int k = 0; k < 4; ++k) v(k) = clone_q[4 * i + k];
    for (int k = 0; k < 3; ++k) v(4 + k) = clone_p[3 * i + k];
    state->_imu->pose()->set_value(v);
    state->_timestamp = 100.0 + 0.1 * i;
    StateHelper::augment_clone(state, w0);
  }
  state->_timestamp = timestamp;
  if (state->max_covariance_size() != N) return -11;
  {
    std::vector<std::shared_ptr<Type>> order;
    order.push_back(state->_imu);
    order.push_back(state->_calib_dt_CAMtoIMU);
    order.push_back(state->_calib_IMUtoCAM.at(0));
    order.push_back(state->_cam_intrinsics.at(0));
    for (auto &c : state->_clones_IMU) order.push_back(c.second);
    MatrixXd Pm(N, N);
    memcpy(Pm.data(), P, sizeof(double) * (size_t)N * N);
    StateHelper::set_initial_covariance(state, Pm, order);
  }
  SimTruth sim;
  for (int k = 0; k < 17; ++k) sim.state_gt[k] = 0.5 + 0.125 * k;
  sim.calib_camimu_dt = 0.0123456;
  for (int k = 0; k < 8; ++k) sim.intrinsics[k] = intr[k] + 1.0;
  for (int k = 0; k < 7; ++k) sim.extrinsics[k] = 0.1 * (k + 1);
  std::ostringstream oe, os, og;
  ROSVisualizerHelper::sim_save_total_state_to_file(state, with_gt ? &sim : nullptr, oe, os, og);
  const std::string a = oe.str(), b = os.str(), c = og.str();
  if ((int)a.size() + 1 > cap || (int)b.size() + 1 > cap || (int)c.size() + 1 > cap) return -1;
  memcpy(est, a.c_str(), a.size() + 1);
  memcpy(sd, b.c_str(), b.size() + 1);
  memcpy(gt, c.c_str(), c.size() + 1);
  return 0;
}

// UpdaterHelper::get_feature_jacobian_full for ONE feature held in landmark representation `rep` anchored in clone slot
// anchor_ci (ext LandmarkRepresentation 0..5).  Outputs column-major: H_f [rows x hf_cols], H_x [rows x cols], res [rows],
// order ids / sizes of the H_x columns.  Needs the device only because a State owns a context.
extern "C" int ovph_feature_jacobian_rep(int C, const double *clone_q, const double *clone_p, const double *clone_q_fej,
                                         const double *clone_p_fej, const double *calib_q, const double *calib_p, const double *intr,
                                         int N, const double *P, int m, const float *uv, const int *clone_idx, const double *p_FinG,
                                         int rep, int anchor_ci, double sigma_px, int do_fej, double *H_f, double *H_x, double *res,
                                         int *rows_out, int *cols_out, int *hf_cols_out, int *order_id, int *order_size, int *n_order) {
  StateOptions so;
  so.do_fej = do_fej != 0;
  so.do_calib_camera_pose = so.do_calib_camera_intrinsics = so.do_calib_camera_timeoffset = true;
  so.max_clone_size = C;
  so.max_state_size = N + 8;
  so.max_features = 16;
  HarnessState hs;
  int rc = build_harness_state(hs, so, C, clone_q, clone_p, clone_q_fej, clone_p_fej, calib_q, calib_p, intr, 0, nullptr, nullptr, 0,
                               nullptr, nullptr, N, P);
  if (rc) return rc;
  auto &state = hs.state;
  UpdaterHelper::UpdaterHelperFeature feat;
  feat.featid = 5000;
  for (int k = 0; k < m; ++k) {
    feat.timestamps.push_back(hs.times[clone_idx[k]]);
    feat.uvs.push_back(uv[2 * k]);
    feat.uvs.push_back(uv[2 * k + 1]);
  }
  memcpy(feat.p_FinG, p_FinG, 3 * sizeof(double));
  memcpy(feat.p_FinG_fej, p_FinG, 3 * sizeof(double));
  feat.feat_representation = (LandmarkRepresentation::Representation)rep;
  if (LandmarkRepresentation::is_relative_representation(feat.feat_representation)) {
    feat.anchor_cam_id = 0;
    feat.anchor_clone_timestamp = hs.times[anchor_ci];
    auto anchor = state->_clones_IMU.at(feat.anchor_clone_timestamp);
    auto calib = state->_calib_IMUtoCAM.at(0);
    const double *Ra = anchor->Rot(), *pa = anchor->pos(), *Rc = calib->Rot(), *pc = calib->pos();
    const double d[3] = {p_FinG[0] - pa[0], p_FinG[1] - pa[1], p_FinG[2] - pa[2]};
    double t[3];
    for (int i = 0; i < 3; ++i) t[i] = Ra[3 * i] * d[0] + Ra[3 * i + 1] * d[1] + Ra[3 * i + 2] * d[2];
    for (int i = 0; i < 3; ++i) feat.p_FinA[i] = Rc[3 * i] * t[0] + Rc[3 * i + 1] * t[1] + Rc[3 * i + 2] * t[2] + pc[i];
    memcpy(feat.p_FinA_fej, feat.p_FinA, 3 * sizeof(double));
  }
  MatrixXd Hf, Hx;
  VectorXd r;
  std::vector<std::shared_ptr<Type>> order;
  UpdaterHelper::get_feature_jacobian_full(state, feat, sigma_px, 1.0, Hf, Hx, r, order);
  *rows_out = Hf.rows();
  *cols_out = Hx.cols();
  *hf_cols_out = Hf.cols();
  memcpy(H_f, Hf.data(), sizeof(double) * (size_t)Hf.rows() * Hf.cols());
  memcpy(H_x, Hx.data(), sizeof(double) * (size_t)Hx.rows() * Hx.cols());
  memcpy(res, r.data(), sizeof(double) * (size_t)r.rows());
  *n_order = (int)order.size();
  for (size_t k = 0; k < order.size(); ++k) {
    order_id[k] = order[k]->id();
    order_size[k] = order[k]->size();
  }
  return 0;
}

// UpdaterSLAM::change_anchors on a state whose first landmark is held in the anchored representation `rep` (2..4) and anchored
// in the oldest clone, with one clone more than max_clone_size in the window: the landmark moves to the newest clone.
// lm_p_FinA / lm_p_FinA_fej: its position in the old anchor camera frame.  Outputs: landmark value / fej (representation
// parameters), its new anchor clone slot, the covariance.
extern "C" int ovph_run_change_anchors(int C, const double *clone_q, const double *clone_p, const double *clone_q_fej,
                                       const double *clone_p_fej, const double *calib_q, const double *calib_p, const double *intr,
                                       int n_slam, const double *slam_p, int N, const double *P, int rep, const double *lm_p_FinA,
                                       const double *lm_p_FinA_fej, int do_fej, double *out_value, double *out_fej, int *out_anchor_ci,
                                       double *out_P) {
  StateOptions so;
  so.do_fej = do_fej != 0;
  so.do_calib_camera_pose = so.do_calib_camera_intrinsics = so.do_calib_camera_timeoffset = true;
  so.max_clone_size = C - 1;
  so.max_state_size = N + 8;
  so.max_features = 16;
  HarnessState hs;
  int rc = build_harness_state(hs, so, C, clone_q, clone_p, clone_q_fej, clone_p_fej, calib_q, calib_p, intr, n_slam, slam_p, slam_p, 0,
                               nullptr, nullptr, N, P);
  if (rc) return rc;
  auto &state = hs.state;
  auto lm = hs.landmarks.at(0);
  lm->_feat_representation = (LandmarkRepresentation::Representation)rep;
  lm->_anchor_cam_id = 0;
  lm->_anchor_clone_timestamp = hs.times[0];
  lm->set_from_xyz(lm_p_FinA, false);
  lm->set_from_xyz(lm_p_FinA_fej, true);
  UpdaterOptions uo, ua;
  ov_core::FeatureInitializerOptions fio;
  UpdaterSLAM up(uo, ua, fio);
  up.change_anchors(state);
  for (int k = 0; k < 3; ++k) {
    out_value[k] = lm->value()(k);
    out_fej[k] = lm->fej()(k);
  }
  *out_anchor_ci = -1;
  for (int i = 0; i < C; ++i)
    if (hs.times[i] == lm->_anchor_clone_timestamp) *out_anchor_ci = i;
  MatrixXd Pn = StateHelper::get_full_covariance(state);
  memcpy(out_P, Pn.data(), sizeof(double) * (size_t)N * N);
  return lm->has_had_anchor_change ? 0 : -20;
}
