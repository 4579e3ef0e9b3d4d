// On-disk formats around the update path: what a run of the reference leaves behind and what is
// needed to compare a run of this build with it offline.
//   - state / std / groundtruth text files      ros/ROSVisualizerHelper.cpp:152-302 (sim_save_total_state_to_file)
//   - timing CSV                                core/VioManager.cpp:110-118 (header), :911-927 (rows)
//   - trajectory input                          data/udel_arl_short.txt: "# timestamp(s) tx ty tz qx qy qz qw"
//   - per-frame binary trace (not in the reference): inputs of one update step (state tables, covariance, feature batch,
//     options) and its outputs (dx, accept mask, chi2, covariance), so that a frame recorded next to the reference
//     (ROS + open_vins required there) can be replayed here and compared.  Layout: see ov_plane_amd/trace.py, the two
//     implementations are tested against each other.
#pragma once
#include <array>
#include <cstdint>
#include <iosfwd>
#include <string>
#include <vector>

#include "ov_plane_host.h"

namespace ov_plane {

// the part of the simulator the groundtruth writer reads (sim/Simulator.h get_state / get_true_parameters)
struct SimTruth {
  double state_gt[17] = {0};    // t, q_GtoI (4), p_IinG (3), v (3), bg (3), ba (3)
  double calib_camimu_dt = 0.0;
  double intrinsics[8] = {0};   // camera 0
  double extrinsics[7] = {0};   // q_ItoC, p_IinC
};

class ROSVisualizerHelper {
public:
  // sim may be null (no groundtruth line).  One line is appended to each stream.
  static void sim_save_total_state_to_file(std::shared_ptr<State> state, const SimTruth *sim, std::ostream &of_state_est,
                                           std::ostream &of_state_std, std::ostream &of_state_gt);
};

// core/VioManager.cpp:110-118 / :911-927
struct TimingRecord {
  double timestamp_inI = 0, track = 0, prop = 0, planeinit = 0, msckf = 0, slam_update = 0, slam_delay = 0, marg = 0, total = 0;
};
void write_timing_header(std::ostream &os, const StateOptions &opts);
void write_timing_row(std::ostream &os, const StateOptions &opts, const TimingRecord &r);

// trajectory file: lines "t tx ty tz qx qy qz qw", '#' comments (sim/Simulator.cpp load_data)
bool load_trajectory(const std::string &path, std::vector<std::array<double, 8>> &poses);

// per-frame trace
struct FrameTrace {
  double timestamp = 0;
  int C = 0, F = 0, M = 0, N = 0;
  std::vector<double> clone_q, clone_p, clone_q_fej, clone_p_fej;
  std::vector<int32_t> clone_id;
  double calib_q[4] = {0, 0, 0, 1}, calib_p[3] = {0, 0, 0}, intrinsics[8] = {0};
  int32_t calib_id = -1, intr_id = -1;
  std::vector<double> P;  // N x N
  std::vector<float> uv;
  std::vector<int32_t> clone_idx, n_meas;
  std::vector<double> p_FinG;
  double sigma_px = 1, chi2_mult = 1, sigma_c = 0.05;
  int32_t do_fej = 1, do_calib_pose = 1, do_calib_intr = 1;
  // outputs
  std::vector<double> dx, chi2, P_after;
  std::vector<uint8_t> accepted;
};
bool write_frame_trace(std::ostream &os, const FrameTrace &f, bool with_header);
bool read_frame_trace(std::istream &is, FrameTrace &f, bool expect_header);
// while a file is open every UpdaterMSCKF::update appends the inputs and outputs of its point update ("" closes it)
bool open_update_trace(const std::string &path);

}  // namespace ov_plane
