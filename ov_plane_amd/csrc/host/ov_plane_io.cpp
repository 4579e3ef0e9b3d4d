#include "ov_plane_io.h"

#include <cmath>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <sstream>

namespace ov_plane {

// ros/ROSVisualizerHelper.cpp:152-302
void ROSVisualizerHelper::sim_save_total_state_to_file(std::shared_ptr<State> state, const SimTruth *sim, std::ostream &of_state_est,
                                                       std::ostream &of_state_std, std::ostream &of_state_gt) {
  const double t_ItoC = state->_calib_dt_CAMtoIMU->value()(0);
  double timestamp_inI = state->_timestamp + t_ItoC;
  if (sim != nullptr) {  // :163-208
    timestamp_inI = state->_timestamp + sim->calib_camimu_dt;
    of_state_gt.precision(5);
    of_state_gt.setf(std::ios::fixed, std::ios::floatfield);
    of_state_gt << sim->state_gt[0] << " ";
    of_state_gt.precision(6);
    for (int k = 1; k < 17; ++k) of_state_gt << sim->state_gt[k] << " ";
    of_state_gt.precision(7);
    of_state_gt << sim->calib_camimu_dt << " ";
    of_state_gt.precision(0);
    of_state_gt << state->_options.num_cameras << " ";
    of_state_gt.precision(6);
    for (int i = 0; i < state->_options.num_cameras; i++) {
      for (int k = 0; k < 8; ++k) of_state_gt << sim->intrinsics[k] << " ";
      for (int k = 0; k < 7; ++k) of_state_gt << sim->extrinsics[k] << " ";
    }
    of_state_gt << std::endl;
  }
  const MatrixXd cov = StateHelper::get_full_covariance(state);  // :215
  auto sd = [&](int id) { return std::sqrt(cov(id, id)); };
  // :218-227
  of_state_est.precision(5);
  of_state_est.setf(std::ios::fixed, std::ios::floatfield);
  of_state_est << timestamp_inI << " ";
  of_state_est.precision(6);
  for (int k = 0; k < 4; ++k) of_state_est << state->_imu->quat()[k] << " ";
  for (int k = 0; k < 3; ++k) of_state_est << state->_imu->pos()[k] << " ";
  for (int k = 0; k < 3; ++k) of_state_est << state->_imu->vel()[k] << " ";
  for (int k = 0; k < 3; ++k) of_state_est << state->_imu->bias_g()[k] << " ";
  for (int k = 0; k < 3; ++k) of_state_est << state->_imu->bias_a()[k] << " ";
  // :230-243
  of_state_std.precision(5);
  of_state_std.setf(std::ios::fixed, std::ios::floatfield);
  of_state_std << timestamp_inI << " ";
  of_state_std.precision(6);
  const int ids[5] = {state->_imu->q()->id(), state->_imu->p()->id(), state->_imu->v()->id(), state->_imu->bg()->id(),
                      state->_imu->ba()->id()};
  for (int b = 0; b < 5; ++b)
    for (int k = 0; k < 3; ++k) of_state_std << sd(ids[b] + k) << " ";
  // :246-260
  of_state_est.precision(7);
  of_state_est << state->_calib_dt_CAMtoIMU->value()(0) << " ";
  of_state_est.precision(0);
  of_state_est << state->_options.num_cameras << " ";
  of_state_est.precision(6);
  if (state->_options.do_calib_camera_timeoffset) of_state_std << sd(state->_calib_dt_CAMtoIMU->id()) << " ";
  else of_state_std << 0.0 << " ";
  of_state_std.precision(0);
  of_state_std << state->_options.num_cameras << " ";
  of_state_std.precision(6);
  // :263-296
  for (int i = 0; i < state->_options.num_cameras; i++) {
    for (int k = 0; k < 8; ++k) of_state_est << state->_cam_intrinsics.at(i)->value()(k) << " ";
    for (int k = 0; k < 7; ++k) of_state_est << state->_calib_IMUtoCAM.at(i)->value()(k) << " ";
    if (state->_options.do_calib_camera_intrinsics) {
      const int index_in = state->_cam_intrinsics.at(i)->id();
      for (int k = 0; k < 8; ++k) of_state_std << sd(index_in + k) << " ";
    } else {
      for (int k = 0; k < 8; ++k) of_state_std << 0.0 << " ";
    }
    if (state->_options.do_calib_camera_pose) {
      const int index_ex = state->_calib_IMUtoCAM.at(i)->id();
      for (int k = 0; k < 6; ++k) of_state_std << sd(index_ex + k) << " ";
    } else {
      for (int k = 0; k < 6; ++k) of_state_std << 0.0 << " ";
    }
  }
  of_state_est << std::endl;
  of_state_std << std::endl;
}

// core/VioManager.cpp:110-118
void write_timing_header(std::ostream &os, const StateOptions &opts) {
  os << "# timestamp (sec),tracking,propagation,";
  if (opts.use_plane_constraint) os << "plane init,";
  os << "msckf update,";
  if (opts.max_slam_features > 0) os << "slam update,slam delayed,";
  os << "re-tri & marg,total" << std::endl;
}

// core/VioManager.cpp:911-927
void write_timing_row(std::ostream &os, const StateOptions &opts, const TimingRecord &r) {
  os << std::fixed << std::setprecision(15) << r.timestamp_inI << "," << std::fixed << std::setprecision(5) << r.track << "," << r.prop
     << ",";
  if (opts.use_plane_constraint) os << r.planeinit << ",";
  os << r.msckf << ",";
  if (opts.max_slam_features > 0) os << r.slam_update << "," << r.slam_delay << ",";
  os << r.marg << "," << r.total << std::endl;
  os.flush();
}

bool load_trajectory(const std::string &path, std::vector<std::array<double, 8>> &poses) {
  std::ifstream file(path);
  if (!file.is_open()) return false;
  std::string line;
  while (std::getline(file, line)) {
    if (line.empty() || line[0] == '#') continue;
    for (auto &ch : line)
      if (ch == ',') ch = ' ';  // both separators occur in open_vins trajectory files
    std::istringstream ss(line);
    std::array<double, 8> v;
    int k = 0;
    while (k < 8 && (ss >> v[k])) ++k;
    if (k != 8) return false;
    poses.push_back(v);
  }
  return !poses.empty();
}

// ---- per-frame trace: little-endian, see ov_plane_amd/trace.py ----
namespace {
const char kMagic[8] = {'O', 'V', 'P', 'T', 'R', 'C', '0', '1'};
template <class T>
void put(std::ostream &os, const T *p, size_t n) {
  os.write(reinterpret_cast<const char *>(p), (std::streamsize)(sizeof(T) * n));
}
template <class T>
bool get(std::istream &is, T *p, size_t n) {
  is.read(reinterpret_cast<char *>(p), (std::streamsize)(sizeof(T) * n));
  return (size_t)is.gcount() == sizeof(T) * n;
}
template <class T>
bool getv(std::istream &is, std::vector<T> &v, size_t n) {
  v.resize(n);
  return n == 0 || get(is, v.data(), n);
}
}  // namespace

bool write_frame_trace(std::ostream &os, const FrameTrace &f, bool with_header) {
  if (with_header) put(os, kMagic, 8);
  const int32_t dims[4] = {f.C, f.F, f.M, f.N};
  put(os, &f.timestamp, 1);
  put(os, dims, 4);
  put(os, f.clone_q.data(), (size_t)4 * f.C);
  put(os, f.clone_p.data(), (size_t)3 * f.C);
  put(os, f.clone_q_fej.data(), (size_t)4 * f.C);
  put(os, f.clone_p_fej.data(), (size_t)3 * f.C);
  put(os, f.clone_id.data(), (size_t)f.C);
  put(os, f.calib_q, 4);
  put(os, f.calib_p, 3);
  put(os, f.intrinsics, 8);
  const int32_t ids[2] = {f.calib_id, f.intr_id};
  put(os, ids, 2);
  put(os, f.P.data(), (size_t)f.N * f.N);
  put(os, f.uv.data(), (size_t)f.F * f.M * 2);
  put(os, f.clone_idx.data(), (size_t)f.F * f.M);
  put(os, f.n_meas.data(), (size_t)f.F);
  put(os, f.p_FinG.data(), (size_t)3 * f.F);
  const double o[3] = {f.sigma_px, f.chi2_mult, f.sigma_c};
  put(os, o, 3);
  const int32_t fl[4] = {f.do_fej, f.do_calib_pose, f.do_calib_intr, (int32_t)(f.dx.empty() ? 0 : 1)};
  put(os, fl, 4);
  if (!f.dx.empty()) {
    put(os, f.dx.data(), (size_t)f.N);
    put(os, f.accepted.data(), (size_t)f.F);
    put(os, f.chi2.data(), (size_t)f.F);
    put(os, f.P_after.data(), (size_t)f.N * f.N);
  }
  return (bool)os;
}

bool read_frame_trace(std::istream &is, FrameTrace &f, bool expect_header) {
  if (expect_header) {
    char m[8];
    if (!get(is, m, 8) || memcmp(m, kMagic, 8) != 0) return false;
  }
  int32_t dims[4];
  if (!get(is, &f.timestamp, 1) || !get(is, dims, 4)) return false;
  f.C = dims[0];
  f.F = dims[1];
  f.M = dims[2];
  f.N = dims[3];
  if (f.C < 0 || f.F < 0 || f.M < 0 || f.N < 0 || f.C > 4096 || f.N > 65536) return false;
  int32_t ids[2], fl[4];
  double o[3];
  bool ok = getv(is, f.clone_q, (size_t)4 * f.C) && getv(is, f.clone_p, (size_t)3 * f.C) && getv(is, f.clone_q_fej, (size_t)4 * f.C) &&
            getv(is, f.clone_p_fej, (size_t)3 * f.C) && getv(is, f.clone_id, (size_t)f.C) && get(is, f.calib_q, 4) && get(is, f.calib_p, 3) &&
            get(is, f.intrinsics, 8) && get(is, ids, 2) && getv(is, f.P, (size_t)f.N * f.N) && getv(is, f.uv, (size_t)f.F * f.M * 2) &&
            getv(is, f.clone_idx, (size_t)f.F * f.M) && getv(is, f.n_meas, (size_t)f.F) && getv(is, f.p_FinG, (size_t)3 * f.F) &&
            get(is, o, 3) && get(is, fl, 4);
  if (!ok) return false;
  f.calib_id = ids[0];
  f.intr_id = ids[1];
  f.sigma_px = o[0];
  f.chi2_mult = o[1];
  f.sigma_c = o[2];
  f.do_fej = fl[0];
  f.do_calib_pose = fl[1];
  f.do_calib_intr = fl[2];
  f.dx.clear();
  f.accepted.clear();
  f.chi2.clear();
  f.P_after.clear();
  if (fl[3])
    ok = getv(is, f.dx, (size_t)f.N) && getv(is, f.accepted, (size_t)f.F) && getv(is, f.chi2, (size_t)f.F) &&
         getv(is, f.P_after, (size_t)f.N * f.N);
  return ok;
}

}  // namespace ov_plane
