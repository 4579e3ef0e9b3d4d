#include "ov_plane_host.h"
#include "ov_plane_io.h"
#include <fstream>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>

using namespace ov_type;

namespace ov_plane {

#define PRINT_ERROR(...) fprintf(stderr, __VA_ARGS__)
#define PRINT_WARNING(...) fprintf(stderr, __VA_ARGS__)
static void gpu_check(int rc, const char *what) {
  if (rc == 0) return;
  PRINT_ERROR("ov_plane(gpu): %s failed: %s\n", what, ovp_error_string(rc));
  std::exit(EXIT_FAILURE);  // the reference treats every failure on this path as fatal (state/StateHelper.cpp:116-118,185-187)
}

// ---- state/State.cpp:33-102 -------------------------------------------------------------------
// The state at construction is a list of (variable, prior standard deviations) entries: the IMU, then - each only when its
// calibration is estimated - the camera time offset and, per camera, extrinsics and intrinsics.  One pass over that list
// assigns the ids, fills _variables and writes the diagonal prior; the variables that are not estimated still exist (the
// measurement model reads their values) but stay out of the covariance (id -1).
State::State(StateOptions &options_) : _options(options_) {
  struct Entry {
    std::shared_ptr<Type> var;
    bool estimated;
    std::vector<double> sigma;  // per error-state component; shorter than size(): the last value repeats
  };
  std::vector<Entry> layout;
  _imu = std::make_shared<IMU>();
  layout.push_back({_imu, true, {1e-3}});                                            // :86 every IMU component 1e-3
  _calib_dt_CAMtoIMU = std::make_shared<Vec>(1);
  layout.push_back({_calib_dt_CAMtoIMU, _options.do_calib_camera_timeoffset, {0.01}});  // :89-91
  for (int cam = 0; cam < _options.num_cameras; ++cam) {
    auto extrinsics = std::make_shared<PoseJPL>();
    auto intrinsics = std::make_shared<Vec>(8);
    _calib_IMUtoCAM.insert({(size_t)cam, extrinsics});
    _cam_intrinsics.insert({(size_t)cam, intrinsics});
    layout.push_back({extrinsics, _options.do_calib_camera_pose, {0.005, 0.005, 0.005, 0.01, 0.01, 0.01}});               // :92-96
    layout.push_back({intrinsics, _options.do_calib_camera_intrinsics, {1.0, 1.0, 1.0, 1.0, 0.005, 0.005, 0.005, 0.005}});  // :97-101
  }
  int n = 0;
  for (Entry &e : layout) {
    if (!e.estimated) continue;
    e.var->set_local_id(n);
    _variables.push_back(e.var);
    n += e.var->size();
  }
  MatrixXd Cov = MatrixXd::Zero(n, n);
  for (const Entry &e : layout) {
    if (!e.estimated) continue;
    for (int k = 0; k < e.var->size(); ++k) {
      const double sd = e.sigma[std::min<size_t>(k, e.sigma.size() - 1)];
      Cov(e.var->id() + k, e.var->id() + k) = sd * sd;
    }
  }
  gpu_check(ovp_ctx_create(_options.gpu_device, _options.max_state_size, std::min(64, _options.max_clone_size + 2), _options.max_features, nullptr, &_gpu),
            "ovp_ctx_create");
  gpu_check(ovp_cov_upload(_gpu, Cov.data(), n, n), "ovp_cov_upload");
  PlaneFitting::bind(_gpu, _options.planefit_shuffle_variant);
}

State::~State() {
  if (PlaneFitting::bound() == _gpu) PlaneFitting::bind(nullptr, 0);
  if (_gpu) ovp_ctx_destroy(_gpu);
}

// ---- state/StateHelper.cpp:41-119 --------------------------------------------------------------
void StateHelper::EKFPropagation(std::shared_ptr<State> state, const std::vector<std::shared_ptr<Type>> &order_NEW,
                                 const std::vector<std::shared_ptr<Type>> &order_OLD, const MatrixXd &Phi, const MatrixXd &Q) {
  if (order_NEW.empty() || order_OLD.empty()) {
    PRINT_ERROR("StateHelper::EKFPropagation() - Called with empty variable arrays!\n");
    std::exit(EXIT_FAILURE);
  }
  int size_order_NEW = order_NEW.at(0)->size();
  for (size_t i = 0; i < order_NEW.size() - 1; i++) {
    if (order_NEW.at(i)->id() + order_NEW.at(i)->size() != order_NEW.at(i + 1)->id()) {
      PRINT_ERROR("StateHelper::EKFPropagation() - Called with non-contiguous state elements!\n");
      std::exit(EXIT_FAILURE);
    }
    size_order_NEW += order_NEW.at(i + 1)->size();
  }
  int size_order_OLD = 0;
  std::vector<int> ids, sizes;
  for (const auto &var : order_OLD) {
    ids.push_back(var->id());
    sizes.push_back(var->size());
    size_order_OLD += var->size();
  }
  assert(size_order_NEW == Phi.rows() && size_order_OLD == Phi.cols());
  assert(size_order_NEW == Q.cols() && size_order_NEW == Q.rows());
  int neg = 0;
  int rc = ovp_cov_propagate(state->_gpu, order_NEW.at(0)->id(), Phi.rows(), ids.data(), sizes.data(), (int)ids.size(), Phi.data(),
                             Q.data(), &neg);
  if (rc == OVP_E_NEGDIAG || neg) {
    PRINT_ERROR("StateHelper::EKFPropagation() - negative covariance diagonal\n");
    std::exit(EXIT_FAILURE);
  }
  gpu_check(rc, "ovp_cov_propagate");
}

// Cholesky of a small SPD host matrix (noise whitening when R != I)
static bool host_llt(MatrixXd &A) {
  const int n = A.rows();
  for (int j = 0; j < n; ++j) {
    double d = A(j, j);
    for (int k = 0; k < j; ++k) d -= A(j, k) * A(j, k);
    if (!(d > 0)) return false;
    d = std::sqrt(d);
    A(j, j) = d;
    for (int i = j + 1; i < n; ++i) {
      double s = A(i, j);
      for (int k = 0; k < j; ++k) s -= A(i, k) * A(j, k);
      A(i, j) = s / d;
    }
  }
  return true;
}

void StateHelper::apply_correction(std::shared_ptr<State> state, const double *dx) {
  for (auto &var : state->_variables) {
    VectorXd d(var->size(), 1);
    for (int k = 0; k < var->size(); ++k) d(k) = dx[var->id() + k];
    var->update(d);
  }
}

// ---- state/StateHelper.cpp:121-202 -------------------------------------------------------------
void StateHelper::EKFUpdate(std::shared_ptr<State> state, const std::vector<std::shared_ptr<Type>> &H_order, const MatrixXd &H,
                            const VectorXd &res, const MatrixXd &R) {
  assert(res.rows() == R.rows());
  assert(H.rows() == res.rows());
  std::vector<int> col_ids;
  for (const auto &v : H_order)
    for (int k = 0; k < v->size(); ++k) col_ids.push_back(v->id() + k);
  assert((int)col_ids.size() == H.cols());
  // the device update assumes whitened noise (R = I, true for every caller on the MSCKF/plane path); whiten otherwise
  bool identity = true;
  for (int j = 0; j < R.cols() && identity; ++j)
    for (int i = 0; i < R.rows(); ++i)
      if (R(i, j) != (i == j ? 1.0 : 0.0)) {
        identity = false;
        break;
      }
  MatrixXd Hw = H;
  VectorXd rw = res;
  if (!identity) {
    MatrixXd Lr = R;
    if (!host_llt(Lr)) {
      PRINT_ERROR("StateHelper::EKFUpdate() - measurement noise is not positive definite\n");
      std::exit(EXIT_FAILURE);
    }
    const int m = H.rows();
    for (int j = 0; j < H.cols(); ++j)
      for (int i = 0; i < m; ++i) {
        double s = Hw(i, j);
        for (int k = 0; k < i; ++k) s -= Lr(i, k) * Hw(k, j);
        Hw(i, j) = s / Lr(i, i);
      }
    for (int i = 0; i < m; ++i) {
      double s = rw(i);
      for (int k = 0; k < i; ++k) s -= Lr(i, k) * rw(k);
      rw(i) = s / Lr(i, i);
    }
  }
  const int n = ovp_cov_size(state->_gpu);
  std::vector<double> dx(n, 0.0);
  ovp_update_info info;
  int rc = ovp_ekf_update(state->_gpu, Hw.data(), Hw.rows(), Hw.cols(), Hw.rows(), col_ids.data(), rw.data(), dx.data(), &info);
  if (rc == OVP_E_NEGDIAG) {
    PRINT_ERROR("StateHelper::EKFUpdate() - negative covariance diagonal\n");
    std::exit(EXIT_FAILURE);
  }
  gpu_check(rc, "ovp_ekf_update");
  apply_correction(state, dx.data());
}

// ---- state/StateHelper.cpp:204-229 -------------------------------------------------------------
void StateHelper::set_initial_covariance(std::shared_ptr<State> state, const MatrixXd &covariance,
                                         const std::vector<std::shared_ptr<Type>> &order) {
  const int n = ovp_cov_size(state->_gpu);
  MatrixXd Cov(n, n);
  gpu_check(ovp_cov_download(state->_gpu, Cov.data(), n, n), "ovp_cov_download");
  int i_index = 0;
  for (size_t i = 0; i < order.size(); i++) {
    int k_index = 0;
    for (size_t k = 0; k < order.size(); k++) {
      for (int a = 0; a < order[i]->size(); ++a)
        for (int b = 0; b < order[k]->size(); ++b) Cov(order[i]->id() + a, order[k]->id() + b) = covariance(i_index + a, k_index + b);
      k_index += order[k]->size();
    }
    i_index += order[i]->size();
  }
  for (int j = 0; j < n; ++j)
    for (int i = j + 1; i < n; ++i) Cov(i, j) = Cov(j, i);  // selfadjointView<Upper>
  gpu_check(ovp_cov_upload(state->_gpu, Cov.data(), n, n), "ovp_cov_upload");
}

// ---- state/StateHelper.cpp:231-274 -------------------------------------------------------------
MatrixXd StateHelper::get_marginal_covariance(std::shared_ptr<State> state, const std::vector<std::shared_ptr<Type>> &small_variables) {
  std::vector<int> ids, sizes;
  int cov_size = 0;
  for (const auto &v : small_variables) {
    ids.push_back(v->id());
    sizes.push_back(v->size());
    cov_size += v->size();
  }
  MatrixXd Small_cov = MatrixXd::Zero(cov_size, cov_size);
  gpu_check(ovp_cov_marginal(state->_gpu, ids.data(), sizes.data(), (int)ids.size(), Small_cov.data()), "ovp_cov_marginal");
  return Small_cov;
}

MatrixXd StateHelper::get_full_covariance(std::shared_ptr<State> state) {
  const int n = ovp_cov_size(state->_gpu);
  MatrixXd full_cov = MatrixXd::Zero(n, n);
  gpu_check(ovp_cov_download(state->_gpu, full_cov.data(), n, n), "ovp_cov_download");
  return full_cov;
}

// ---- state/StateHelper.cpp:276-344 -------------------------------------------------------------
// The covariance loses the variable's rows / columns on the device; on the host the variable leaves _variables and everything
// behind it moves up by its size.
void StateHelper::marginalize(std::shared_ptr<State> state, std::shared_ptr<Type> marg) {
  auto &vars = state->_variables;
  const auto where = std::find(vars.begin(), vars.end(), marg);
  if (where == vars.end()) {
    PRINT_ERROR("StateHelper::marginalize() - the variable is not part of the state\n");
    std::exit(EXIT_FAILURE);
  }
  const int gone_at = marg->id(), gone = marg->size();
  gpu_check(ovp_cov_marginalize(state->_gpu, gone_at, gone), "ovp_cov_marginalize");
  vars.erase(where);
  for (auto &v : vars)
    if (v->id() > gone_at) v->set_local_id(v->id() - gone);
  marg->set_local_id(-1);
}

// ---- state/StateHelper.cpp:346-396 -------------------------------------------------------------
// The variable to copy is either a state variable itself or a sub-variable of one (the IMU's pose): the first state variable that
// owns it decides where its block sits; the copy goes to the end of the covariance.
std::shared_ptr<Type> StateHelper::clone(std::shared_ptr<State> state, std::shared_ptr<Type> variable_to_clone) {
  std::shared_ptr<Type> source;
  for (const auto &v : state->_variables) {
    source = (v == variable_to_clone) ? v : v->check_if_subvariable(variable_to_clone);
    if (source == variable_to_clone) break;
    source = nullptr;
  }
  if (!source) {
    PRINT_ERROR("StateHelper::clone() - the variable is not part of the state\n");
    std::exit(EXIT_FAILURE);
  }
  const int appended_at = ovp_cov_size(state->_gpu);
  gpu_check(ovp_cov_clone(state->_gpu, source->id(), variable_to_clone->size()), "ovp_cov_clone");
  std::shared_ptr<Type> copy = source->clone();
  copy->set_local_id(appended_at);
  state->_variables.push_back(copy);
  return copy;
}

// ---- state/StateHelper.cpp:588-625 -------------------------------------------------------------
// Stochastic clone of the IMU pose at the state's time; with the camera time offset in the state the clone also depends on it
// through the motion during the offset (d clone / d dt = [w; v], :613-624).
void StateHelper::augment_clone(std::shared_ptr<State> state, const double last_w[3]) {
  const double now = state->_timestamp;
  if (state->_clones_IMU.count(now)) {
    PRINT_ERROR("StateHelper::augment_clone() - there is a clone at this time already\n");
    std::exit(EXIT_FAILURE);
  }
  auto pose = std::dynamic_pointer_cast<PoseJPL>(StateHelper::clone(state, state->_imu->pose()));
  if (!pose) {
    PRINT_ERROR("StateHelper::augment_clone() - the clone of the IMU pose is not a pose\n");
    std::exit(EXIT_FAILURE);
  }
  state->_clones_IMU[now] = pose;
  if (!state->_options.do_calib_camera_timeoffset) return;
  const double *v = state->_imu->vel();
  const double dnc_dt[6] = {last_w[0], last_w[1], last_w[2], v[0], v[1], v[2]};
  gpu_check(ovp_cov_augment_dt(state->_gpu, pose->id(), state->_calib_dt_CAMtoIMU->id(), dnc_dt), "ovp_cov_augment_dt");
}

// ---- state/StateHelper.cpp:627-636 -------------------------------------------------------------
// One clone more than the window holds: the oldest goes (std::map keeps the clones ordered by time).
void StateHelper::marginalize_old_clone(std::shared_ptr<State> state) {
  auto &clones = state->_clones_IMU;
  if ((int)clones.size() <= state->_options.max_clone_size) return;
  const auto oldest = clones.begin();
  StateHelper::marginalize(state, oldest->second);
  clones.erase(oldest);
}

// ---- state/StateHelper.cpp:638-652 -------------------------------------------------------------
// Landmarks flagged by the updaters leave the state (ArUco ids, the first 4 * max_aruco_features, are never dropped).
void StateHelper::marginalize_slam(std::shared_ptr<State> state) {
  std::vector<size_t> leaving;
  for (const auto &lm : state->_features_SLAM)
    if (lm.second->should_marg && (int)lm.first > 4 * state->_options.max_aruco_features) leaving.push_back(lm.first);
  for (size_t id : leaving) {
    StateHelper::marginalize(state, state->_features_SLAM.at(id));
    state->_features_SLAM_to_PLANE.erase(id);
    state->_features_SLAM.erase(id);
  }
}

// ---- state/StateHelper.cpp:654-776 -------------------------------------------------------------
void StateHelper::merge_planes_and_marginalize(std::shared_ptr<State> state, const std::map<size_t, size_t> &feat2plane,
                                               const std::map<size_t, std::set<size_t>> &plane2oldplane) {
  if (state->_features_PLANE.empty()) return;
  auto it5 = state->_features_PLANE.begin();
  while (it5 != state->_features_PLANE.end()) {
    const size_t planeid = (*it5).first;
    int planeid_new = -1;
    bool in_state = false;
    for (auto const &planeset : plane2oldplane) {
      if (planeset.second.find(planeid) != planeset.second.end()) {
        planeid_new = (int)planeset.first;
        in_state = (state->_features_PLANE.find(planeset.first) != state->_features_PLANE.end());
      }
    }
    if (planeid_new == -1 || (int)planeid == planeid_new) {
      it5++;
      continue;
    }
    if (!in_state) {  // the surviving id is not a state variable: the old variable simply takes the new id
      state->_features_PLANE.insert({(size_t)planeid_new, state->_features_PLANE.at(planeid)});
      it5 = state->_features_PLANE.erase(it5);
      continue;
    }
    auto plane_new = state->_features_PLANE.at((size_t)planeid_new);
    auto plane_old = state->_features_PLANE.at(planeid);
    double cn[3], co[3], nn = 0, no = 0, dot = 0;
    for (int k = 0; k < 3; ++k) {
      cn[k] = plane_new->value()(k);
      co[k] = plane_old->value()(k);
      nn += cn[k] * cn[k];
      no += co[k] * co[k];
    }
    nn = std::sqrt(nn);
    no = std::sqrt(no);
    for (int k = 0; k < 3; ++k) dot += (cn[k] / nn) * (co[k] / no);
    const double norm_angle = (180.0 / M_PI) * std::acos(dot);
    // 3 rows: cp_new - cp_old = 0 with sigma_plane_merge
    const double white_c = 1.0 / state->_options.sigma_plane_merge;
    VectorXd res(3, 1);
    MatrixXd H = MatrixXd::Zero(3, 6);
    for (int k = 0; k < 3; ++k) {
      res(k) = white_c * (0.0 - (cn[k] - co[k]));
      H(k, k) = white_c;
      H(k, 3 + k) = -white_c;
    }
    std::vector<std::shared_ptr<Type>> H_order = {plane_new, plane_old};
    MatrixXd R = MatrixXd::Identity(3, 3);
    // S = H P_marg H^T + R is 3x3: (Pnn - Pno - Pon + Poo) / sigma^2 + I
    MatrixXd P_marg = StateHelper::get_marginal_covariance(state, H_order);
    double S[9], L[9] = {0};
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j)
        S[3 * i + j] = white_c * white_c * (P_marg(i, j) - P_marg(i, 3 + j) - P_marg(3 + i, j) + P_marg(3 + i, 3 + j)) + (i == j ? 1.0 : 0.0);
    for (int j = 0; j < 3; ++j) {  // LLT
      double d = S[3 * j + j];
      for (int k = 0; k < j; ++k) d -= L[3 * j + k] * L[3 * j + k];
      L[3 * j + j] = std::sqrt(d);
      for (int i = j + 1; i < 3; ++i) {
        double s = S[3 * i + j];
        for (int k = 0; k < j; ++k) s -= L[3 * i + k] * L[3 * j + k];
        L[3 * i + j] = s / L[3 * j + j];
      }
    }
    double y[3], chi2 = 0.0;
    for (int i = 0; i < 3; ++i) {
      double s = res(i);
      for (int k = 0; k < i; ++k) s -= L[3 * i + k] * y[k];
      y[i] = s / L[3 * i + i];
      chi2 += y[i] * y[i];
    }
    const double chi2_check = state->_options.plane_merge_chi2 * ovp_chi2_quantile_095(3);
    if (chi2 < chi2_check && norm_angle < state->_options.plane_merge_deg_max) StateHelper::EKFUpdate(state, H_order, H, res, R);
    StateHelper::marginalize(state, plane_old);
    it5 = state->_features_PLANE.erase(it5);
  }
  // planes without any observing feature leave the state
  std::set<size_t> active_planes;
  for (auto const &featpair : feat2plane) active_planes.insert(featpair.second);
  it5 = state->_features_PLANE.begin();
  while (it5 != state->_features_PLANE.end()) {
    if (active_planes.find((*it5).first) == active_planes.end()) {
      StateHelper::marginalize(state, (*it5).second);
      it5 = state->_features_PLANE.erase(it5);
    } else {
      it5++;
    }
  }
}

// ---- Eigen pieces used by initialize ----------------------------------------
static void make_givens(double p, double q, double &c, double &s) {
  if (q == 0.0) {
    c = p < 0 ? -1.0 : 1.0;
    s = 0.0;
  } else if (p == 0.0) {
    c = 0.0;
    s = q < 0 ? 1.0 : -1.0;
  } else if (std::fabs(p) > std::fabs(q)) {
    double t = q / p, u = std::sqrt(1.0 + t * t);
    if (p < 0) u = -u;
    c = 1.0 / u;
    s = -t * c;
  } else {
    double t = p / q, u = std::sqrt(1.0 + t * t);
    if (q < 0) u = -u;
    s = -1.0 / u;
    c = -t * s;
  }
}
static void rot_rows(MatrixXd &A, int r0, int c0, double c, double s) {
  for (int j = c0; j < A.cols(); ++j) {
    const double x = A(r0, j), y = A(r0 + 1, j);
    A(r0, j) = c * x - s * y;
    A(r0 + 1, j) = s * x + c * y;
  }
}
static void check_isotropic(const MatrixXd &R, const char *who) {
  assert(R.rows() == R.cols());
  assert(R.rows() > 0);
  for (int r = 0; r < R.rows(); r++)
    for (int c = 0; c < R.cols(); c++) {
      if (r == c && R(0, 0) != R(r, c)) {
        PRINT_ERROR("StateHelper::%s() - Your noise is not isotropic!\n", who);
        std::exit(EXIT_FAILURE);
      } else if (r != c && R(r, c) != 0.0) {
        PRINT_ERROR("StateHelper::%s() - Your noise is not diagonal!\n", who);
        std::exit(EXIT_FAILURE);
      }
    }
}
static bool small_inverse(const MatrixXd &A, MatrixXd &Ainv) {  // replaces colPivHouseholderQr().inverse() (:564)
  const int k = A.rows();
  MatrixXd M(k, 2 * k);
  for (int i = 0; i < k; ++i)
    for (int j = 0; j < k; ++j) {
      M(i, j) = A(i, j);
      M(i, k + j) = (i == j) ? 1.0 : 0.0;
    }
  for (int c = 0; c < k; ++c) {
    int piv = c;
    for (int r = c + 1; r < k; ++r)
      if (std::fabs(M(r, c)) > std::fabs(M(piv, c))) piv = r;
    if (M(piv, c) == 0.0) return false;
    if (piv != c)
      for (int j = 0; j < 2 * k; ++j) std::swap(M(c, j), M(piv, j));
    const double d = M(c, c);
    for (int j = 0; j < 2 * k; ++j) M(c, j) /= d;
    for (int r = 0; r < k; ++r)
      if (r != c) {
        const double f = M(r, c);
        for (int j = 0; j < 2 * k; ++j) M(r, j) -= f * M(c, j);
      }
  }
  Ainv.resize(k, k);
  for (int i = 0; i < k; ++i)
    for (int j = 0; j < k; ++j) Ainv(i, j) = M(i, k + j);
  return true;
}

// OVP_HOST_INIT_SPLIT=1 keeps the three separate device calls (gate / augmentation / update) for A/B timing and tests
static bool fused_initialize_enabled() {
  const char *e = getenv("OVP_HOST_INIT_SPLIT");  // read per call: the tests flip it inside one process
  return !(e && e[0] == '1');
}

// ---- state/StateHelper.cpp:398-487 -------------------------------------------------------------
bool StateHelper::initialize(std::shared_ptr<State> state, std::shared_ptr<Type> new_variable,
                             const std::vector<std::shared_ptr<Type>> &H_order, MatrixXd &H_R, MatrixXd &H_L, MatrixXd &R,
                             VectorXd &res, double chi_2_mult, bool do_update) {
  if (std::find(state->_variables.begin(), state->_variables.end(), new_variable) != state->_variables.end()) {
    PRINT_ERROR("StateHelper::initialize_invertible() - Called on variable that is already in the state\n");
    std::exit(EXIT_FAILURE);
  }
  check_isotropic(R, "initialize");
  const int new_var_size = new_variable->size();
  assert(new_var_size == H_L.cols());
  // :434-446 Givens split
  for (int n = 0; n < H_L.cols(); ++n)
    for (int m = H_L.rows() - 1; m > n; m--) {
      double c, s;
      make_givens(H_L(m - 1, n), H_L(m, n), c, s);
      rot_rows(H_L, m - 1, n, c, s);
      rot_rows(res, m - 1, 0, c, s);
      rot_rows(H_R, m - 1, 0, c, s);
    }
  const int rows = H_R.rows(), cols = H_R.cols(), rup = rows - new_var_size;
  MatrixXd Hxinit = H_R.block(0, 0, new_var_size, cols);
  MatrixXd H_finit = H_L.block(0, 0, new_var_size, new_var_size);
  VectorXd resinit = res.block(0, 0, new_var_size, 1);
  MatrixXd Rinit = R.block(0, 0, new_var_size, new_var_size);
  MatrixXd Hup = H_R.block(new_var_size, 0, rup, cols);
  VectorXd resup = res.block(new_var_size, 0, rup, 1);
  MatrixXd Rup = R.block(new_var_size, new_var_size, rup, rup);
  // Device path: gate (:464-475), initialize_invertible (:477-480) and the update with the remaining rows (:483-485) as one
  // enqueue with one synchronisation (ovp_cov_initialize); OVP_E_CAPACITY = outside that entry's limits, nothing was touched.
  if (fused_initialize_enabled()) {
    std::vector<int> col_ids;
    for (const auto &v : H_order)
      for (int k = 0; k < v->size(); ++k) col_ids.push_back(v->id() + k);
    MatrixXd H_Linv;
    if (!small_inverse(H_finit, H_Linv)) {
      PRINT_ERROR("StateHelper::initialize() - H_L is singular\n");
      std::exit(EXIT_FAILURE);
    }
    const int oldSize = ovp_cov_size(state->_gpu);
    std::vector<double> dx((size_t)oldSize + new_var_size, 0.0);
    int accepted = 0;
    double chi2_dev = 0.0;
    const double thr = chi_2_mult * ovp_chi2_quantile_095(res.rows());
    int rc = ovp_cov_initialize(state->_gpu, Hxinit.data(), rup > 0 ? Hup.data() : nullptr, new_var_size, rup, cols, col_ids.data(),
                                H_Linv.data(), Rinit.data(), rup > 0 ? resup.data() : nullptr, rup > 0 ? Rup(0, 0) : 1.0, thr,
                                do_update ? 1 : 0, &accepted, &chi2_dev, dx.data());
    if (rc != OVP_E_CAPACITY) {
      if (rc == OVP_E_NEGDIAG) {
        PRINT_ERROR("StateHelper::EKFUpdate() - negative covariance diagonal\n");
        std::exit(EXIT_FAILURE);
      }
      gpu_check(rc, "ovp_cov_initialize");
      if (!accepted) return false;
      VectorXd d(new_var_size, 1);  // :577 new_variable->update(H_Linv * res)
      for (int i = 0; i < new_var_size; ++i) {
        double sacc = 0.0;
        for (int a = 0; a < new_var_size; ++a) sacc += H_Linv(i, a) * resinit(a);
        d(i) = sacc;
      }
      new_variable->update(d);
      new_variable->set_local_id(oldSize);
      state->_variables.push_back(new_variable);
      if (rup > 0 && do_update) apply_correction(state, dx.data());
      return true;
    }
  }
  // :464-475 Mahalanobis test of the update part against the prior, dof = res.rows()
  double chi2 = 0.0;
  if (rup > 0) {
    MatrixXd P_up = get_marginal_covariance(state, H_order);
    // S = Hup P_up Hup^T + R as column axpys on the raw column-major storage (the compiler vectorises these; the products are
    // 2 rup cols^2 flops, the largest host-side cost of a delayed initialisation at 30 clones)
    MatrixXd HP(rup, cols);
    {
      const double *__restrict hp = Hup.data();
      double *__restrict out = HP.data();
      for (int b = 0; b < cols; ++b) {
        double *__restrict ob = out + (size_t)b * rup;
        for (int a = 0; a < cols; ++a) {
          const double pv = P_up(a, b);
          if (pv == 0.0) continue;
          const double *__restrict ha = hp + (size_t)a * rup;
          for (int i = 0; i < rup; ++i) ob[i] += ha[i] * pv;
        }
      }
    }
    MatrixXd S = Rup;
    {
      const double *__restrict hpv = HP.data();
      const double *__restrict hu = Hup.data();
      double *__restrict sp = S.data();
      for (int a = 0; a < cols; ++a) {
        const double *__restrict ca = hpv + (size_t)a * rup;
        for (int j = 0; j < rup; ++j) {
          const double hv = hu[(size_t)a * rup + j];
          if (hv == 0.0) continue;
          double *__restrict sj = sp + (size_t)j * rup;
          for (int i = 0; i < rup; ++i) sj[i] += ca[i] * hv;
        }
      }
    }
    if (!host_llt(S)) return false;
    VectorXd tmp = resup;
    for (int i = 0; i < rup; ++i) {
      double s = tmp(i);
      for (int k = 0; k < i; ++k) s -= S(i, k) * tmp(k);
      tmp(i) = s / S(i, i);
    }
    for (int i = 0; i < rup; ++i) chi2 += tmp(i) * tmp(i);  // res^T S^-1 res = |L^-1 res|^2
  }
  const double chi2_check = ovp_chi2_quantile_095(res.rows());
  if (chi2 > chi_2_mult * chi2_check) return false;
  StateHelper::initialize_invertible(state, new_variable, H_order, Hxinit, H_finit, Rinit, resinit);
  if (Hup.rows() > 0 && do_update) StateHelper::EKFUpdate(state, H_order, Hup, resup, Rup);
  return true;
}

// ---- state/StateHelper.cpp:489-586 -------------------------------------------------------------
void StateHelper::initialize_invertible(std::shared_ptr<State> state, std::shared_ptr<Type> new_variable,
                                        const std::vector<std::shared_ptr<Type>> &H_order, const MatrixXd &H_R, const MatrixXd &H_L,
                                        const MatrixXd &R, const VectorXd &res) {
  if (std::find(state->_variables.begin(), state->_variables.end(), new_variable) != state->_variables.end()) {
    PRINT_ERROR("StateHelper::initialize_invertible() - Called on variable that is already in the state\n");
    std::exit(EXIT_FAILURE);
  }
  check_isotropic(R, "initialize_invertible");
  assert(res.rows() == R.rows());
  assert(H_L.rows() == res.rows());
  assert(H_L.rows() == H_R.rows());
  assert(H_L.rows() == H_L.cols());
  assert(H_L.rows() == new_variable->size());
  std::vector<int> col_ids;
  for (const auto &v : H_order)
    for (int k = 0; k < v->size(); ++k) col_ids.push_back(v->id() + k);
  MatrixXd H_Linv;
  if (!small_inverse(H_L, H_Linv)) {
    PRINT_ERROR("StateHelper::initialize_invertible() - H_L is singular\n");
    std::exit(EXIT_FAILURE);
  }
  const int oldSize = ovp_cov_size(state->_gpu);
  gpu_check(ovp_cov_initialize_invertible(state->_gpu, H_R.data(), H_R.rows(), H_R.cols(), H_R.rows(), col_ids.data(), H_Linv.data(),
                                          R.data()),
            "ovp_cov_initialize_invertible");
  // :577 new_variable->update(H_Linv * res)
  VectorXd d(new_variable->size(), 1);
  for (int i = 0; i < new_variable->size(); ++i) {
    double s = 0.0;
    for (int a = 0; a < H_Linv.cols(); ++a) s += H_Linv(i, a) * res(a);
    d(i) = s;
  }
  new_variable->update(d);
  new_variable->set_local_id(oldSize);
  state->_variables.push_back(new_variable);
}

// ---- update/UpdaterMSCKF.cpp ---------------------------------------------------------------------
UpdaterMSCKF::UpdaterMSCKF(UpdaterOptions &options, ov_core::FeatureInitializerOptions &feat_init_options)
    : _options(options), _featinit(feat_init_options) {
  _options.sigma_pix_sq = std::pow(_options.sigma_pix, 2);
  // the chi-square table (:59-62) lives inside libovplane_hip.so (ovp_chi2_quantile_095)
}

// ---- per-frame trace of the point update: every UpdaterMSCKF::update appends one record while a file is open
static std::ofstream g_update_trace;
static bool g_update_trace_header = false;
bool open_update_trace(const std::string &path) {
  if (g_update_trace.is_open()) g_update_trace.close();
  if (path.empty()) return true;
  g_update_trace.open(path, std::ios::binary);
  g_update_trace_header = true;
  return g_update_trace.is_open();
}
static std::ostream *update_trace_stream() { return g_update_trace.is_open() ? &g_update_trace : nullptr; }
static bool update_trace_take_header() {
  const bool h = g_update_trace_header;
  g_update_trace_header = false;
  return h;
}

void UpdaterMSCKF::update(std::shared_ptr<State> state, std::vector<std::shared_ptr<ov_core::Feature>> &feature_vec,
                          std::vector<std::shared_ptr<ov_core::Feature>> &feature_vec_extra,
                          std::vector<std::shared_ptr<ov_core::Feature>> &feature_vec_used, const std::map<size_t, size_t> &feat2plane) {
  if (feature_vec.empty()) return;  // :70-71

  // :74-100  keep measurements at existing clone times, drop features with < 2 of them
  std::map<double, int> clone_slot;
  std::vector<std::shared_ptr<PoseJPL>> clones;
  for (const auto &c : state->_clones_IMU) {
    clone_slot[c.first] = (int)clones.size();
    clones.push_back(c.second);
  }
  auto it0 = feature_vec.begin();
  while (it0 != feature_vec.end()) {
    auto &ft = **it0;
    std::vector<float> uv2, uvn2;
    std::vector<double> ts2;
    std::vector<int> cam2;
    const bool has_norm = ft.uvs_norm.size() == ft.uvs.size();
    for (size_t k = 0; k < ft.timestamps.size(); ++k)
      if (clone_slot.count(ft.timestamps[k])) {
        ts2.push_back(ft.timestamps[k]);
        if (!ft.cam_ids.empty()) cam2.push_back(ft.cam_of(k));
        uv2.push_back(ft.uvs[2 * k]);
        uv2.push_back(ft.uvs[2 * k + 1]);
        if (has_norm) {
          uvn2.push_back(ft.uvs_norm[2 * k]);
          uvn2.push_back(ft.uvs_norm[2 * k + 1]);
        }
      }
    ft.timestamps = ts2;
    ft.cam_ids = cam2;
    ft.uvs = uv2;
    ft.uvs_norm = uvn2;
    if (ts2.size() < 2) {
      ft.to_delete = true;
      it0 = feature_vec.erase(it0);
    } else {
      it0++;
    }
  }
  if (feature_vec.empty()) return;

  // ---- pack the pose tables and the feature batch for the device ----
  const int C = (int)clones.size();
  std::vector<double> cq(4 * C), cp(3 * C), cqf(4 * C), cpf(3 * C);
  std::vector<int> cid(C);
  for (int i = 0; i < C; ++i) {
    memcpy(&cq[4 * i], clones[i]->quat(), 4 * sizeof(double));
    memcpy(&cp[3 * i], clones[i]->pos(), 3 * sizeof(double));
    memcpy(&cqf[4 * i], clones[i]->quat_fej(), 4 * sizeof(double));
    memcpy(&cpf[3 * i], clones[i]->pos_fej(), 3 * sizeof(double));
    cid[i] = clones[i]->id();
  }
  auto pack_tables = [&]() {
    ovp_state_tables st;
    st.n_state = ovp_cov_size(state->_gpu);
    st.n_clones = C;
    for (int i = 0; i < C; ++i) {  // values may have changed after the plane loop
      memcpy(&cq[4 * i], clones[i]->quat(), 4 * sizeof(double));
      memcpy(&cp[3 * i], clones[i]->pos(), 3 * sizeof(double));
    }
    st.clone_q = cq.data();
    st.clone_p = cp.data();
    st.clone_q_fej = cqf.data();
    st.clone_p_fej = cpf.data();
    st.clone_id = cid.data();
    auto calib = state->_calib_IMUtoCAM.at(0);
    auto intr = state->_cam_intrinsics.at(0);
    memcpy(st.calib_q, calib->quat(), 4 * sizeof(double));
    memcpy(st.calib_p, calib->pos(), 3 * sizeof(double));
    st.calib_id = calib->id();
    memcpy(st.intrinsics, intr->value().data(), 8 * sizeof(double));
    st.intr_id = intr->id();
    st.cam_fisheye = (state->_cam_fisheye.count(0) && state->_cam_fisheye.at(0)) ? 1 : 0;
    gpu_check(ovp_state_upload(state->_gpu, &st), "ovp_state_upload");
  };
  // what the device batch can carry: camera 0's measurements, at most OVP_MAX_MEAS of them (one wavefront's rows).  Everything
  // else takes the dense side channel of the point update below (ovp_msckf_dense_blocks); in the uploads in front of it (device
  // triangulation, plane loop) such a feature is present with NO measurements, i.e. it takes no part there
  auto fits_batch = [](const ov_core::Feature &f) { return f.only_camera0() && (int)f.timestamps.size() <= OVP_MAX_MEAS; };
  // measurements of a feature that go into a batch upload: all of them when the feature fits; for the device TRIANGULATION a
  // feature that does not fit still takes part with its first OVP_MAX_MEAS measurements of camera 0 (a position estimate needs
  // no more; the update below linearises over all of them); for the update's batches it has none
  auto batch_meas = [&](const ov_core::Feature &f, bool for_triangulation) {
    std::vector<size_t> sel;
    if (fits_batch(f)) {
      for (size_t k = 0; k < f.timestamps.size(); ++k) sel.push_back(k);
    } else if (for_triangulation) {
      for (size_t k = 0; k < f.timestamps.size() && (int)sel.size() < OVP_MAX_MEAS; ++k)
        if (f.cam_of(k) == 0) sel.push_back(k);
    }
    return sel;
  };
  auto upload_batch = [&](const std::vector<std::shared_ptr<ov_core::Feature>> &fv, bool for_triangulation = false) {
    const int F = (int)fv.size();
    int M = 1;
    for (auto &f : fv) M = std::max(M, (int)batch_meas(*f, for_triangulation).size());
    std::vector<float> uv((size_t)F * M * 2, 0.f);
    std::vector<int> cidx((size_t)F * M, -1), nm(F);
    std::vector<double> pf((size_t)F * 3);
    for (int f = 0; f < F; ++f) {
      const std::vector<size_t> sel = batch_meas(*fv[f], for_triangulation);
      nm[f] = (int)sel.size();
      for (int k = 0; k < nm[f]; ++k) {
        cidx[(size_t)f * M + k] = clone_slot.at(fv[f]->timestamps[sel[k]]);
        uv[((size_t)f * M + k) * 2] = fv[f]->uvs[2 * sel[k]];
        uv[((size_t)f * M + k) * 2 + 1] = fv[f]->uvs[2 * sel[k] + 1];
      }
      memcpy(&pf[3 * f], fv[f]->p_FinG, 3 * sizeof(double));
    }
    ovp_feature_batch fb{F, M, uv.data(), cidx.data(), nm.data(), pf.data()};
    gpu_check(ovp_batch_upload(state->_gpu, &fb), "ovp_batch_upload");
    return M;
  };
  ovp_update_opts o{_options.sigma_pix,
                    _options.chi2_multipler,
                    state->_options.sigma_constraint,
                    state->_options.do_fej ? 1 : 0,
                    state->_options.do_calib_camera_pose ? 1 : 0,
                    state->_options.do_calib_camera_intrinsics ? 1 : 0,
                    0};
  pack_tables();

  // ---- :120-166 triangulate (+ refine) the features that arrive with normalised measurements; failures are erased ----
  {
    bool any_norm = false;
    for (auto &f : feature_vec) any_norm = any_norm || (!f->uvs_norm.empty() && f->uvs_norm.size() == f->uvs.size());
    if (any_norm) {
      const int M = upload_batch(feature_vec, true);
      const int F = (int)feature_vec.size();
      std::vector<float> uvn((size_t)F * M * 2, 0.f);
      for (int f = 0; f < F; ++f) {
        if (feature_vec[f]->uvs_norm.size() != feature_vec[f]->uvs.size()) continue;  // (position handed over)
        const std::vector<size_t> sel = batch_meas(*feature_vec[f], true);
        for (size_t k = 0; k < sel.size(); ++k) {
          uvn[((size_t)f * M + k) * 2] = feature_vec[f]->uvs_norm[2 * sel[k]];
          uvn[((size_t)f * M + k) * 2 + 1] = feature_vec[f]->uvs_norm[2 * sel[k] + 1];
        }
      }
      ovp_triang_opts to;
      to.refine_features = _featinit.refine_features ? 1 : 0;
      to.triangulate_1d = _featinit.triangulate_1d ? 1 : 0;
      to.reserved = 0;
      to.max_runs = _featinit.max_runs;
      to.init_lamda = _featinit.init_lamda;
      to.max_lamda = _featinit.max_lamda;
      to.min_dx = _featinit.min_dx;
      to.min_dcost = _featinit.min_dcost;
      to.lam_mult = _featinit.lam_mult;
      to.min_dist = _featinit.min_dist;
      to.max_dist = _featinit.max_dist;
      to.max_baseline = _featinit.max_baseline;
      to.max_cond_number = _featinit.max_cond_number;
      std::vector<double> pf((size_t)F * 3);
      std::vector<uint8_t> okv(F, 0);
      gpu_check(ovp_triangulate(state->_gpu, &to, uvn.data(), pf.data(), okv.data()), "ovp_triangulate");
      size_t f = 0;
      auto it1 = feature_vec.begin();
      while (it1 != feature_vec.end()) {
        const bool had_norm = !(*it1)->uvs_norm.empty();
        if (had_norm && !okv[f]) {
          (*it1)->to_delete = true;  // :161-165
          it1 = feature_vec.erase(it1);
        } else {
          if (had_norm) memcpy((*it1)->p_FinG, &pf[3 * f], 3 * sizeof(double));
          it1++;
        }
        ++f;
      }
      if (feature_vec.empty()) return;
    }
  }

  // ---- :196-400 plane linearisation points ----
  // Planes in the state: their features are refined against the (fixed) plane.  Other planes: RANSAC fit of the triangulated
  // points, then joint refinement of plane and points; a plane that fails either step is skipped this frame.  Runs when the
  // on-plane features carry normalised measurements (the tracker's uvs_norm) and the caller has not handed estimates over in
  // state->_plane_estimates_cp_inG (the pre-fitted entry point used when the fit happens elsewhere).
  std::map<size_t, std::vector<double>> plane_estimates = state->_plane_estimates_cp_inG;
  std::set<size_t> fitted_planes;       // planes whose feature set went through the fit below
  std::set<size_t> plane_feat_kept;     // their surviving features (plane_feats.at(planeid) of :421)
  std::map<size_t, std::vector<size_t>> plane_slam_kept;  // SLAM landmarks among them (planes outside the state only)
  if (state->_options.use_plane_constraint && state->_options.use_plane_constraint_msckf && !feat2plane.empty() &&
      state->_plane_estimates_cp_inG.empty()) {
    std::map<size_t, std::vector<std::shared_ptr<ov_core::Feature>>> plane_feats;  // :198
    bool all_norm = true;
    auto collect = [&](std::vector<std::shared_ptr<ov_core::Feature>> &vec) {  // :206-228
      for (auto &feat : vec) {
        auto it = feat2plane.find(feat->featid);
        if (it == feat2plane.end()) continue;
        if (!fits_batch(*feat)) continue;  // (the fit / refinement kernels know camera 0 and 32 views: such a feature stays a point feature)
        all_norm = all_norm && (feat->uvs_norm.size() == 2 * feat->timestamps.size());
        plane_feats[it->second].push_back(feat);
      }
    };
    collect(feature_vec);
    {  // :101-118 the extra features get the same clean-up as the main vector
      auto itx = feature_vec_extra.begin();
      while (itx != feature_vec_extra.end()) {
        auto &ft = **itx;
        std::vector<float> uv2, uvn2;
        std::vector<double> ts2;
        std::vector<int> cam2;
        const bool has_norm = ft.uvs_norm.size() == ft.uvs.size();
        for (size_t k = 0; k < ft.timestamps.size(); ++k)
          if (clone_slot.count(ft.timestamps[k])) {
            ts2.push_back(ft.timestamps[k]);
            if (!ft.cam_ids.empty()) cam2.push_back(ft.cam_of(k));
            uv2.push_back(ft.uvs[2 * k]);
            uv2.push_back(ft.uvs[2 * k + 1]);
            if (has_norm) {
              uvn2.push_back(ft.uvs_norm[2 * k]);
              uvn2.push_back(ft.uvs_norm[2 * k + 1]);
            }
          }
        ft.timestamps = ts2;
        ft.cam_ids = cam2;
        ft.uvs = uv2;
        ft.uvs_norm = uvn2;
        if (ts2.size() < 2) itx = feature_vec_extra.erase(itx);
        else itx++;
      }
    }
    collect(feature_vec_extra);
    if (all_norm && !plane_feats.empty()) {
      // :232-252 SLAM features of planes that are not in the state take part in the fit as constants
      if (state->_options.use_plane_constraint_slamu) {
        for (auto &lm : state->_features_SLAM) {
          auto it = feat2plane.find(lm.first);
          if (it == feat2plane.end() || state->_features_PLANE.count(it->second)) continue;
          auto bad = state->_features_SLAM_to_PLANE.find(lm.first);
          if (bad != state->_features_SLAM_to_PLANE.end() && bad->second == 0) continue;
          auto fp = std::make_shared<ov_core::Feature>();
          fp->featid = lm.first;
          lm.second->get_xyz(false, fp->p_FinG);
          plane_feats[it->second].push_back(fp);
        }
      }
      // camera poses of the clones (:122-141): R_GtoCi = R_ItoC R_GtoIi, p_CiinG = p_IiinG - R_GtoCi^T p_IinC
      PlaneFitting::ClonesCam clones_cam;
      auto calib = state->_calib_IMUtoCAM.at(0);
      for (const auto &cl : state->_clones_IMU) {
        PlaneFitting::ClonePose cpose;
        const double *Ri = cl.second->Rot(), *Rc = calib->Rot(), *pi = cl.second->pos(), *pc = calib->pos();
        for (int i = 0; i < 3; ++i)
          for (int k = 0; k < 3; ++k) cpose.R[3 * i + k] = Rc[3 * i] * Ri[k] + Rc[3 * i + 1] * Ri[3 + k] + Rc[3 * i + 2] * Ri[6 + k];
        for (int i = 0; i < 3; ++i)
          cpose.p[i] = pi[i] - (cpose.R[i] * pc[0] + cpose.R[3 + i] * pc[1] + cpose.R[6 + i] * pc[2]);
        clones_cam[0][cl.first] = cpose;
      }
      const double focal_length = state->_cam_intrinsics.at(0)->value()(0);  // :269-272
      const double sigma_px_norm = _options.sigma_pix / focal_length;
      const double sigma_c = state->_options.sigma_constraint;
      double stateI[7], calib0[7];
      memcpy(stateI, state->_imu->quat(), 4 * sizeof(double));
      memcpy(stateI + 4, state->_imu->pos(), 3 * sizeof(double));
      memcpy(calib0, calib->quat(), 4 * sizeof(double));
      memcpy(calib0 + 4, calib->pos(), 3 * sizeof(double));
      for (auto &fp : plane_feats) {  // :262-401, std::map order
        const size_t pid = fp.first;
        fitted_planes.insert(pid);
        auto &feats = fp.second;
        double cp[3];
        if (state->_features_PLANE.count(pid)) {  // :265-316
          auto pl = state->_features_PLANE.at(pid);
          for (int a = 0; a < 3; ++a) cp[a] = pl->value()(a);
          if (state->_options.use_refine_plane_feat &&
              !PlaneFitting::optimize_plane(feats, cp, clones_cam, sigma_px_norm, sigma_c, true, stateI, calib0))
            continue;
          // :284-302 ground truth for the features (the plane itself is a state variable and keeps its estimate)
          if (state->_options.use_groundtruths && !state->_true_planes.empty() && !state->_true_features.empty())
            for (auto &ft : feats) {
              auto itt = state->_true_features.find(ft->featid);
              if (itt != state->_true_features.end()) memcpy(ft->p_FinG, itt->second.data(), 3 * sizeof(double));
            }
        } else {
          if (feats.size() < 4) continue;  // :320-321
          double abcd[4];
          if (!PlaneFitting::plane_fitting(feats, abcd, state->_options.plane_msckf_min_feat, state->_options.plane_msckf_max_cond))
            continue;  // :325-327
          for (int a = 0; a < 3; ++a) cp[a] = -abcd[a] * abcd[3];  // :352
          if (state->_options.use_refine_plane_feat &&
              !PlaneFitting::optimize_plane(feats, cp, clones_cam, sigma_px_norm, sigma_c, false, stateI, calib0))
            continue;  // :355-357
          // :363-380 ground truth for the plane and its features
          if (state->_options.use_groundtruths && !state->_true_planes.empty() && !state->_true_features.empty()) {
            auto itp = state->_true_planes.find(pid);
            if (itp != state->_true_planes.end()) memcpy(cp, itp->second.data(), 3 * sizeof(double));
            for (auto &ft : feats) {
              auto itt = state->_true_features.find(ft->featid);
              if (itt != state->_true_features.end()) memcpy(ft->p_FinG, itt->second.data(), 3 * sizeof(double));
            }
          }
          bool has_msckf_feat = false;  // :384-392
          for (auto &ft : feats) has_msckf_feat = has_msckf_feat || !state->_features_SLAM.count(ft->featid);
          if (!has_msckf_feat || feats.size() < 4) continue;  // :395-396
        }
        plane_estimates[pid] = {cp[0], cp[1], cp[2]};
        for (auto &ft : feats) {
          plane_feat_kept.insert(ft->featid);
          if (ft->timestamps.empty() && state->_features_SLAM.count(ft->featid)) plane_slam_kept[pid].push_back(ft->featid);
        }
      }
      // on-plane extra features that survived join the batch of the plane loop (plane_feats.at(planeid), :421)
      for (auto &ft : feature_vec_extra)
        if (plane_feat_kept.count(ft->featid) && plane_estimates.count(feat2plane.at(ft->featid))) feature_vec.push_back(ft);
    }
  }

  // ---- plane loop (:411-649) ----
  std::set<size_t> features_used_already;
  if (state->_options.use_plane_constraint && state->_options.use_plane_constraint_msckf && !feat2plane.empty()) {
    // planes that have an estimate: in the state (when no fit ran), fitted above, or handed over by the caller
    std::vector<size_t> plane_ids;
    for (const auto &fp : feat2plane)
      if (std::find(plane_ids.begin(), plane_ids.end(), fp.second) == plane_ids.end()) plane_ids.push_back(fp.second);
    std::sort(plane_ids.begin(), plane_ids.end());  // std::map iteration order of plane_estimates_cp_inG (:413)
    std::vector<size_t> used_planes;
    for (size_t pid : plane_ids) {
      const bool in_state = state->_features_PLANE.count(pid) > 0;
      if (fitted_planes.empty() ? (in_state || plane_estimates.count(pid)) : plane_estimates.count(pid) > 0) used_planes.push_back(pid);
    }
    if (!used_planes.empty()) {
      const int NP = (int)used_planes.size();
      std::vector<int> pof(feature_vec.size(), 0), sid(NP, -1);
      std::vector<double> cpv(3 * NP), cpfej(3 * NP);
      for (int k = 0; k < NP; ++k) {
        const size_t pid = used_planes[k];
        if (state->_features_PLANE.count(pid)) {
          auto pl = state->_features_PLANE.at(pid);
          sid[k] = pl->id();
          for (int a = 0; a < 3; ++a) {
            cpv[3 * k + a] = pl->value()(a);
            cpfej[3 * k + a] = pl->fej()(a);
          }
        } else {
          for (int a = 0; a < 3; ++a) cpv[3 * k + a] = cpfej[3 * k + a] = plane_estimates.at(pid)[a];
        }
      }
      for (size_t f = 0; f < feature_vec.size(); ++f) {
        auto it = feat2plane.find(feature_vec[f]->featid);
        if (it == feat2plane.end()) continue;
        if (fitted_planes.count(it->second) && !plane_feat_kept.count(feature_vec[f]->featid)) continue;  // not an inlier of the fit
        // the plane kernels take a feature's 2 m bearing rows in one wavefront (the constraint row is wave-uniform): a track with
        // more than 32 observations keeps its bearing measurements but not the plane constraint (it goes through the point loop)
        if (feature_vec[f]->timestamps.size() > 32 || !feature_vec[f]->only_camera0()) continue;
        auto pos = std::find(used_planes.begin(), used_planes.end(), it->second);
        if (pos != used_planes.end()) pof[f] = 1 + (int)(pos - used_planes.begin());
      }
      upload_batch(feature_vec);
      const int n = ovp_cov_size(state->_gpu);
      std::vector<double> dxp((size_t)NP * n, 0.0);
      std::vector<uint8_t> pok(NP, 0), fused(feature_vec.size(), 0);
      ovp_plane_batch pb{NP, pof.data(), cpv.data(), cpfej.data(), sid.data()};
      // SLAM landmarks lying on planes that are not in the state: one constraint row each inside that plane's update (:454-552)
      std::vector<int> s_plane, s_id;
      std::vector<size_t> s_featid;
      std::vector<double> s_p, s_pf;
      for (int k = 0; k < NP; ++k) {
        auto itp = plane_slam_kept.find(used_planes[k]);
        if (itp == plane_slam_kept.end() || sid[k] >= 0) continue;
        int n_on_plane = 0;
        for (size_t fid : itp->second) {
          // the device loop takes OVP_PLANE_MAX_SLAM landmark rows per plane; a more crowded plane keeps the first ones (the
          // others stay plain SLAM landmarks of this frame) instead of losing every plane constraint of the frame
          if (++n_on_plane > OVP_PLANE_MAX_SLAM) {
            PRINT_WARNING("UpdaterMSCKF::update() - plane %zu: more than %d SLAM landmarks, the rest is not constrained\n",
                          used_planes[k], OVP_PLANE_MAX_SLAM);
            break;
          }
          auto lm = state->_features_SLAM.at(fid);
          double v[3], vf[3];
          lm->get_xyz(false, v);
          lm->get_xyz(true, vf);
          s_plane.push_back(k + 1);
          s_id.push_back(lm->id());
          s_featid.push_back(fid);
          s_p.insert(s_p.end(), v, v + 3);
          s_pf.insert(s_pf.end(), vf, vf + 3);
        }
      }
      pb.n_slam = (int)s_id.size();
      pb.slam_plane = s_plane.data();
      pb.slam_state_id = s_id.data();
      pb.slam_p = s_p.data();
      pb.slam_p_fej = s_pf.data();
      std::vector<int> pdof(NP, 0);
      {
        const int rcp = ovp_msckf_plane_update(state->_gpu, &o, &pb, dxp.data(), pok.data(), nullptr, pdof.data(), fused.data());
        if (rcp == OVP_E_CAPACITY) {
          // the planes of this frame involve more than 287 columns (clones + calibration + in-state planes + landmarks on the
          // others; the state itself may be larger): the regularities are not used in this update, every feature takes the
          // point loop - the filter degrades to plain MSCKF instead of stopping
          PRINT_ERROR("UpdaterMSCKF::update() - the planes of this frame involve too many columns (state: %d), skipped\n", n);
          std::fill(pok.begin(), pok.end(), 0);
          std::fill(fused.begin(), fused.end(), 0);
        } else {
          gpu_check(rcp, "ovp_msckf_plane_update");
        }
      }
      for (int k = 0; k < NP; ++k)
        if (pok[k]) StateHelper::apply_correction(state, &dxp[(size_t)k * n]);  // :648 per accepted plane, in order
      for (size_t q = 0; q < s_id.size(); ++q) {  // :626-639
        const int k = s_plane[q] - 1;
        if (pok[k]) state->_features_SLAM_to_PLANE[s_featid[q]] = used_planes[k];
        else if (pdof[k] > 0) state->_features_SLAM_to_PLANE[s_featid[q]] = 0;  // the plane ran and failed its chi2 test
      }
      for (size_t f = 0; f < feature_vec.size(); ++f)
        if (fused[f]) {  // :640-644
          feature_vec[f]->to_delete = true;
          features_used_already.insert(feature_vec[f]->featid);
          feature_vec_used.push_back(feature_vec[f]);
        }
    }
  }

  // :657-668 remove features already used
  std::vector<std::shared_ptr<ov_core::Feature>> feature_vec_tmp;
  for (auto const &feature : feature_vec)
    if (features_used_already.find(feature->featid) == features_used_already.end()) feature_vec_tmp.push_back(feature);
  feature_vec = feature_vec_tmp;
  if (feature_vec.empty()) return;

  // ---- point loop, gate, compression, update (:671-814) ----
  if (!features_used_already.empty()) pack_tables();  // the plane loop moved the linearisation points
  upload_batch(feature_vec);
  const int n = ovp_cov_size(state->_gpu);
  std::vector<double> dx(n, 0.0);
  std::vector<uint8_t> ok(feature_vec.size(), 0);
  ovp_update_info info;
  // optional per-frame trace (ov_plane_io.h FrameTrace): everything this update reads, and below what it returned
  FrameTrace tr;
  std::ostream *tos = update_trace_stream();
  if (tos) {
    const int F = (int)feature_vec.size();
    int M = 1;
    for (auto &f : feature_vec)
      if (fits_batch(*f)) M = std::max(M, (int)f->timestamps.size());
    tr.timestamp = state->_timestamp;
    tr.C = C;
    tr.F = F;
    tr.M = M;
    tr.N = n;
    for (int i = 0; i < C; ++i) {
      memcpy(&cq[4 * i], clones[i]->quat(), 4 * sizeof(double));
      memcpy(&cp[3 * i], clones[i]->pos(), 3 * sizeof(double));
    }
    tr.clone_q = cq;
    tr.clone_p = cp;
    tr.clone_q_fej = cqf;
    tr.clone_p_fej = cpf;
    tr.clone_id.assign(cid.begin(), cid.end());
    auto calib = state->_calib_IMUtoCAM.at(0);
    auto intr = state->_cam_intrinsics.at(0);
    memcpy(tr.calib_q, calib->quat(), 4 * sizeof(double));
    memcpy(tr.calib_p, calib->pos(), 3 * sizeof(double));
    memcpy(tr.intrinsics, intr->value().data(), 8 * sizeof(double));
    tr.calib_id = state->_options.do_calib_camera_pose ? calib->id() : -1;
    tr.intr_id = state->_options.do_calib_camera_intrinsics ? intr->id() : -1;
    tr.P.assign((size_t)n * n, 0.0);
    gpu_check(ovp_cov_download(state->_gpu, tr.P.data(), n, n), "ovp_cov_download");
    tr.uv.assign((size_t)F * M * 2, 0.f);
    tr.clone_idx.assign((size_t)F * M, -1);
    tr.n_meas.assign(F, 0);
    tr.p_FinG.assign((size_t)F * 3, 0.0);
    for (int f = 0; f < F; ++f) {
      tr.n_meas[f] = fits_batch(*feature_vec[f]) ? (int)feature_vec[f]->timestamps.size() : 0;
      for (int k = 0; k < tr.n_meas[f]; ++k) {
        tr.clone_idx[(size_t)f * M + k] = clone_slot.at(feature_vec[f]->timestamps[k]);
        tr.uv[((size_t)f * M + k) * 2] = feature_vec[f]->uvs[2 * k];
        tr.uv[((size_t)f * M + k) * 2 + 1] = feature_vec[f]->uvs[2 * k + 1];
      }
      memcpy(&tr.p_FinG[3 * f], feature_vec[f]->p_FinG, 3 * sizeof(double));
    }
    tr.sigma_px = o.sigma_px;
    tr.chi2_mult = o.chi2_multiplier;
    tr.sigma_c = o.sigma_constraint;
    tr.do_fej = o.do_fej;
    tr.do_calib_pose = o.do_calib_camera_pose;
    tr.do_calib_intr = o.do_calib_camera_intrinsics;
  }
  std::vector<double> chi2v(tos ? feature_vec.size() : 0, 0.0);
  // ---- features the batch format cannot carry (a second camera's measurements, a track longer than OVP_MAX_MEAS): the reference
  // treats them like any other (:695-786 - Jacobian over every camera's measurements, nullspace projection, gate); here each
  // becomes a dense block beside the batch and joins the same EKF update (ovp_msckf_dense_blocks) ----
  std::vector<size_t> dense_idx;
  for (size_t f = 0; f < feature_vec.size(); ++f)
    if (!fits_batch(*feature_vec[f])) dense_idx.push_back(f);
  std::vector<uint8_t> dense_ok(dense_idx.size(), 0);
  if (!dense_idx.empty()) {
    std::vector<int> b_rows, b_cols, b_ids;
    std::vector<double> b_H, b_res;
    for (size_t f : dense_idx) {
      const ov_core::Feature &ft = *feature_vec[f];
      UpdaterHelper::UpdaterHelperFeature hf;
      hf.featid = ft.featid;
      hf.uvs = ft.uvs;
      hf.timestamps = ft.timestamps;
      hf.cam_ids = ft.cam_ids;
      hf.feat_representation = LandmarkRepresentation::GLOBAL_3D;  // (the projected system of an MSCKF feature does not depend on it)
      memcpy(hf.p_FinG, ft.p_FinG, 3 * sizeof(double));
      memcpy(hf.p_FinG_fej, ft.p_FinG, 3 * sizeof(double));  // :721-722
      MatrixXd H_f, H_x;
      VectorXd r;
      std::vector<std::shared_ptr<Type>> order;
      UpdaterHelper::get_feature_jacobian_full(state, hf, _options.sigma_pix, state->_options.sigma_constraint, H_f, H_x, r, order);
      UpdaterHelper::nullspace_project_inplace(H_f, H_x, r);  // :730
      b_rows.push_back(H_x.rows());
      b_cols.push_back(H_x.cols());
      b_H.insert(b_H.end(), H_x.data(), H_x.data() + (size_t)H_x.rows() * H_x.cols());
      b_res.insert(b_res.end(), r.data(), r.data() + r.rows());
      for (const auto &v : order)
        for (int k = 0; k < v->size(); ++k) b_ids.push_back(v->id() + k);
    }
    gpu_check(ovp_msckf_dense_blocks(state->_gpu, _options.chi2_multipler, (int)dense_idx.size(), b_rows.data(), b_cols.data(), b_H.data(),
                                     b_ids.data(), b_res.data(), dense_ok.data(), nullptr),
              "ovp_msckf_dense_blocks");
  }
  int rc;
  if (_comm || _world > 1) {
    // feature-sharded point loop: the batch above is the same on every replica; accepted / chi2 come back for this rank's share
    // and are completed over the communicator (errors are collective, see ovplane_hip.h)
    rc = ovp_msckf_update_sharded(state->_gpu, &o, _comm, _rank, _world, dx.data(), ok.data(), tos ? chi2v.data() : nullptr, &info,
                                  &_shard_lo, &_shard_hi);
    if (rc == 0 || rc == OVP_E_NEGDIAG) {
      const int rg = ovp_rccl_gather_decisions(state->_gpu, _comm, ok.data(), tos ? chi2v.data() : nullptr);
      if (rc == 0) rc = rg;
      info.n_accepted = 0;
      for (uint8_t a : ok) info.n_accepted += a ? 1 : 0;
    }
  } else {
    rc = ovp_msckf_update(state->_gpu, &o, dx.data(), ok.data(), tos ? chi2v.data() : nullptr, &info);
  }
  if (rc == OVP_E_NEGDIAG) {
    PRINT_ERROR("StateHelper::EKFUpdate() - negative covariance diagonal\n");
    std::exit(EXIT_FAILURE);
  }
  gpu_check(rc, "ovp_msckf_update");
  for (size_t i = 0; i < dense_idx.size(); ++i) {  // the dense blocks' gate decisions take their places in the vector's order
    ok[dense_idx[i]] = dense_ok[i];
    info.n_accepted += dense_ok[i] ? 1 : 0;
  }
  if (tos) {
    tr.dx = dx;
    tr.accepted = ok;
    tr.chi2 = chi2v;
    tr.P_after.assign((size_t)n * n, 0.0);
    gpu_check(ovp_cov_download(state->_gpu, tr.P_after.data(), n, n), "ovp_cov_download");
    write_frame_trace(*tos, tr, update_trace_take_header());
  }
  // :755-757 rejected features are flagged and erased, :791-793 the rest is flagged as used
  std::vector<std::shared_ptr<ov_core::Feature>> kept;
  for (size_t f = 0; f < feature_vec.size(); ++f) {
    feature_vec[f]->to_delete = true;
    if (ok[f]) kept.push_back(feature_vec[f]);
  }
  feature_vec = kept;
  if (info.n_accepted > 0) StateHelper::apply_correction(state, dx.data());
}

}  // namespace ov_plane
