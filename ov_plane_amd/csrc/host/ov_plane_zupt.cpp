// Host mirror of ov_plane::UpdaterZeroVelocity (update/UpdaterZeroVelocity.cpp).  Built MI355X-first like the other
// updaters: the covariance never leaves the device; the detector reads a 9 x 9 marginal (StateHelper::get_marginal_covariance),
// the bias random walk is a StateHelper::EKFPropagation and the stacked IMU rows are one StateHelper::EKFUpdate, both on the
// device.  The reference hard-codes integrated_accel_constraint = false, model_time_varying_bias = true,
// override_with_disparity_check = true, explicitly_enforce_zero_motion = false (:113-116); the branches those constants
// switch off are not built.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "ov_plane_host.h"
#include "ovplane_hip.h"

#define PRINT_WARNING(...) fprintf(stderr, __VA_ARGS__)

using namespace ov_type;

namespace ov_plane {

UpdaterZeroVelocity::UpdaterZeroVelocity(UpdaterOptions &options, NoiseManager &noises, std::shared_ptr<ov_core::FeatureDatabase> db,
                                         std::shared_ptr<Propagator> prop, double gravity_mag, double zupt_max_velocity,
                                         double zupt_noise_multiplier, double zupt_max_disparity)
    : _options(options), _noises(noises.with_squares()), _db(db), _prop(prop), _gravity{0.0, 0.0, gravity_mag},
      _zupt_max_velocity(zupt_max_velocity), _zupt_noise_multiplier(zupt_noise_multiplier), _zupt_max_disparity(zupt_max_disparity) {
  // (update/UpdaterZeroVelocity.cpp:60-65: the chi2 table is ovp_chi2_quantile_095 of the C-ABI, shared by all updaters)
}

void UpdaterZeroVelocity::feed_imu(const ov_core::ImuData &message, double oldest_time) { imu_buffer_push(imu_data, message, oldest_time); }

bool UpdaterZeroVelocity::try_update(std::shared_ptr<State> state, double timestamp) {
  // Every way out that is not an accepted update forgets the previous zero-velocity time (update/UpdaterZeroVelocity.cpp:71-111,
  // :232, :256): one exit for all of them
  auto no_update = [this]() {
    last_zupt_state_timestamp = 0.0;
    return false;
  };
  // nothing to test with, or the state is already at this image
  if (imu_data.empty() || state->_timestamp == timestamp) return no_update();
  // IMU window [state time, image time], both shifted by the camera time offset as it was when each end was stamped (:83-101):
  // the offset in force at the previous call opens the window, the current estimate closes it and is remembered for the next call
  const double dt_cam_now = state->_calib_dt_CAMtoIMU->value()(0);
  const double dt_cam_before = have_last_prop_time_offset ? last_prop_time_offset : dt_cam_now;
  have_last_prop_time_offset = true;
  last_prop_time_offset = dt_cam_now;
  const std::vector<ov_core::ImuData> imu_recent =
      Propagator::select_imu_readings(imu_data, state->_timestamp + dt_cam_before, timestamp + dt_cam_now);
  if (imu_recent.size() < 2) {
    PRINT_WARNING("[ZUPT]: fewer than two IMU readings between the state and the image: no zero-velocity test\n");
    return no_update();
  }
  // :119-125 order [q, bg, ba]
  const std::vector<std::shared_ptr<Type>> Hx_order{state->_imu->q(), state->_imu->bg(), state->_imu->ba()};
  const int h_size = 9, n_int = (int)imu_recent.size() - 1, m_size = 6 * n_int;
  MatrixXd H = MatrixXd::Zero(m_size, h_size);
  VectorXd res = VectorXd::Zero(m_size, 1);
  MatrixXd R = MatrixXd::Identity(m_size, m_size);
  // :141-176 w_true = w_m - bw - nw = 0 ;  a_true = a_m - ba - R g - na = 0
  const double *Rv = state->_imu->Rot();
  const double *Rj = state->_options.do_fej ? state->_imu->Rot_fej() : state->_imu->Rot();
  double Rg[3], Rjg[3];
  for (int i = 0; i < 3; ++i) {
    Rg[i] = Rv[3 * i] * _gravity[0] + Rv[3 * i + 1] * _gravity[1] + Rv[3 * i + 2] * _gravity[2];
    Rjg[i] = Rj[3 * i] * _gravity[0] + Rj[3 * i + 1] * _gravity[1] + Rj[3 * i + 2] * _gravity[2];
  }
  const double S3[9] = {0, -Rjg[2], Rjg[1], Rjg[2], 0, -Rjg[0], -Rjg[1], Rjg[0], 0};
  double dt_summed = 0;
  for (int i = 0; i < n_int; i++) {
    const double dt = imu_recent[i + 1].timestamp - imu_recent[i].timestamp;
    for (int k = 0; k < 3; ++k) {
      const double a_hat = imu_recent[i].am[k] - state->_imu->bias_a()[k];
      res(6 * i + k) = -(imu_recent[i].wm[k] - state->_imu->bias_g()[k]);
      res(6 * i + 3 + k) = -(a_hat - Rg[k]);
      H(6 * i + k, 3 + k) = -1.0;
      for (int j = 0; j < 3; ++j) H(6 * i + 3 + k, j) = -S3[3 * k + j];
      H(6 * i + 3 + k, 6 + k) = -1.0;
      R(6 * i + k, 6 * i + k) *= _noises.sigma_w_2 / dt;  // continuous -> discrete, :168-174
      R(6 * i + 3 + k, 6 * i + 3 + k) *= _noises.sigma_a_2 / dt;
    }
    dt_summed += dt;
  }
  for (int i = 0; i < m_size; ++i) R(i, i) *= _zupt_noise_multiplier;  // :180
  // :184-186 bias random walk over the window (sigma, not sigma^2, as the reference has it)
  MatrixXd Q_bias = MatrixXd::Identity(6, 6);
  for (int k = 0; k < 3; ++k) {
    Q_bias(k, k) *= dt_summed * _noises.sigma_wb;
    Q_bias(3 + k, 3 + k) *= dt_summed * _noises.sigma_ab;
  }
  // :191-194 chi2 with the propagation "we would do before the update"
  MatrixXd P_marg = StateHelper::get_marginal_covariance(state, Hx_order);
  for (int a = 0; a < 6; ++a)
    for (int b = 0; b < 6; ++b) P_marg(3 + a, 3 + b) += Q_bias(a, b);
  MatrixXd HP = MatrixXd::Zero(m_size, h_size);
  for (int b = 0; b < h_size; ++b)
    for (int a = 0; a < h_size; ++a) {
      const double pv = P_marg(a, b);
      for (int i = 0; i < m_size; ++i) HP(i, b) += H(i, a) * pv;
    }
  MatrixXd S = R;
  for (int a = 0; a < h_size; ++a)
    for (int j = 0; j < m_size; ++j) {
      const double hv = H(j, a);
      if (hv == 0.0) continue;
      for (int i = 0; i < m_size; ++i) S(i, j) += HP(i, a) * hv;
    }
  // res^T S^-1 res = |L^-1 res|^2 (S.llt().solve)
  double chi2 = 0.0;
  {
    bool spd = true;
    for (int j = 0; j < m_size && spd; ++j) {
      double d = S(j, j);
      for (int k = 0; k < j; ++k) d -= S(j, k) * S(j, k);
      if (!(d > 0.0)) {
        spd = false;
        break;
      }
      d = std::sqrt(d);
      S(j, j) = d;
      for (int i = j + 1; i < m_size; ++i) {
        double s = S(i, j);
        for (int k = 0; k < j; ++k) s -= S(i, k) * S(j, k);
        S(i, j) = s / d;
      }
    }
    if (!spd) return no_update();
    VectorXd y = res;
    for (int i = 0; i < m_size; ++i) {
      double s = y(i);
      for (int k = 0; k < i; ++k) s -= S(i, k) * y(k);
      y(i) = s / S(i, i);
      chi2 += y(i) * y(i);
    }
  }
  _last_chi2 = chi2;
  const double chi2_check = ovp_chi2_quantile_095(m_size);  // :197-204
  // :207-227 disparity of the tracks between the last state time and this image
  bool disparity_passed = false;
  {
    int num_features = 0;
    double average_disparity = 0.0, variance_disparity = 0.0;
    if (_db) ov_core::FeatureHelper::compute_disparity(_db, state->_timestamp, timestamp, average_disparity, variance_disparity, num_features);
    disparity_passed = (average_disparity < _zupt_max_disparity && num_features > 20);
  }
  const double *v = state->_imu->vel();
  const double vnorm = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  if (!disparity_passed && (chi2 > _options.chi2_multipler * chi2_check || vnorm > _zupt_max_velocity)) return no_update();  // :231-236
  // :245-247 we will not clone at this time: drop the measurements taken at the previous zero-velocity time
  if (last_zupt_state_timestamp > 0.0 && _db) _db->cleanup_measurements_exact(last_zupt_state_timestamp);
  // :256-262 bias random walk, Phi = I
  {
    const std::vector<std::shared_ptr<Type>> biases{state->_imu->bg(), state->_imu->ba()};
    StateHelper::EKFPropagation(state, biases, biases, MatrixXd::Identity(6, 6), Q_bias);
  }
  StateHelper::EKFUpdate(state, Hx_order, H, res, R);  // :265
  state->_timestamp = timestamp;
  last_zupt_state_timestamp = timestamp;
  return true;
}

}  // namespace ov_plane
