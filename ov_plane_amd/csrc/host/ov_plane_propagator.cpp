// Host mirror of ov_plane::Propagator (state/Propagator.cpp).  The per-interval work is a handful of 3x3 / 15x15 products
// (about 40 intervals per camera frame), so it stays on the host exactly as SURVEY.md §8 a11 scopes it; the only part that
// touches the covariance is StateHelper::EKFPropagation + augment_clone, which run on the device-resident P.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "ov_plane_host.h"

#define PRINT_ERROR(...) fprintf(stderr, __VA_ARGS__)
#define PRINT_WARNING(...) fprintf(stderr, __VA_ARGS__)

using namespace ov_type;

namespace ov_plane {

namespace {
// ---- ext quat_ops.h on row-major 3x3 arrays ---------------------------------
struct M3 {
  double a[9];
  double &operator()(int i, int j) { return a[3 * i + j]; }
  double operator()(int i, int j) const { return a[3 * i + j]; }
};
M3 eye3() { return M3{{1, 0, 0, 0, 1, 0, 0, 0, 1}}; }
M3 from(const double *r) {
  M3 m;
  memcpy(m.a, r, sizeof(m.a));
  return m;
}
M3 mul(const M3 &A, const M3 &B) {
  M3 C;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C(i, j) = A(i, 0) * B(0, j) + A(i, 1) * B(1, j) + A(i, 2) * B(2, j);
  return C;
}
M3 tr(const M3 &A) {
  M3 T;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) T(i, j) = A(j, i);
  return T;
}
M3 skew(const double w[3]) { return M3{{0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0}}; }
double norm3(const double w[3]) { return std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]); }

M3 exp_so3(const double w[3]) {
  const M3 S = skew(w), S2 = mul(S, S);
  const double th = norm3(w);
  const double A = th < 1e-7 ? 1.0 : std::sin(th) / th;
  const double B = th < 1e-7 ? 0.5 : (1.0 - std::cos(th)) / (th * th);
  M3 R = eye3();
  for (int i = 0; i < 9; ++i) R.a[i] += A * S.a[i] + B * S2.a[i];
  return R;
}
M3 Jl_so3(const double w[3]) {
  const double th = norm3(w);
  if (th < 1e-6) return eye3();
  const double a[3] = {w[0] / th, w[1] / th, w[2] / th};
  const M3 S = skew(a);
  const double s = std::sin(th) / th, c = (1.0 - std::cos(th)) / th;
  M3 J;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) J(i, j) = (i == j ? s : 0.0) + (1.0 - s) * a[i] * a[j] + c * S(i, j);
  return J;
}
M3 Jr_so3(const double w[3]) {
  const double m[3] = {-w[0], -w[1], -w[2]};
  return Jl_so3(m);
}
void Omega_times(const double w[3], const double q[4], double o[4]) {  // Omega(w) q
  o[0] = w[2] * q[1] - w[1] * q[2] + w[0] * q[3];
  o[1] = -w[2] * q[0] + w[0] * q[2] + w[1] * q[3];
  o[2] = w[1] * q[0] - w[0] * q[1] + w[2] * q[3];
  o[3] = -w[0] * q[0] - w[1] * q[1] - w[2] * q[2];
}
void quatnorm(double q[4]) {
  if (q[3] < 0)
    for (int k = 0; k < 4; ++k) q[k] = -q[k];
  const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int k = 0; k < 4; ++k) q[k] /= n;
}
void tmul(const M3 &R, const double v[3], double o[3]) {  // R^T v
  for (int i = 0; i < 3; ++i) o[i] = R(0, i) * v[0] + R(1, i) * v[1] + R(2, i) * v[2];
}

// column-major 15x15 helpers
struct M15 {
  double *d;
  double &operator()(int i, int j) { return d[15 * j + i]; }
};
void put(double *M, int r0, int c0, const M3 &B, double s) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) M[15 * (c0 + j) + r0 + i] = s * B(i, j);
}
void mm15(const double *A, const double *B, double *C) {  // C = A B
  for (int j = 0; j < 15; ++j)
    for (int i = 0; i < 15; ++i) {
      double s = 0.0;
      for (int k = 0; k < 15; ++k) s += A[15 * k + i] * B[15 * j + k];
      C[15 * j + i] = s;
    }
}
void mmt15(const double *A, const double *B, double *C) {  // C = A B^T
  for (int j = 0; j < 15; ++j)
    for (int i = 0; i < 15; ++i) {
      double s = 0.0;
      for (int k = 0; k < 15; ++k) s += A[15 * k + i] * B[15 * k + j];
      C[15 * j + i] = s;
    }
}
}  // namespace

Propagator::Propagator(NoiseManager noises, double gravity_mag)
    : _noises(noises.with_squares()), _gravity{0.0, 0.0, gravity_mag}, _Phi{}, _Qs{}, _last_w{} {}

void Propagator::feed_imu(const ov_core::ImuData &message, double oldest_time) {
  std::lock_guard<std::mutex> lck(imu_data_mtx);
  imu_buffer_push(imu_data, message, oldest_time);
}

ov_core::ImuData Propagator::interpolate_data(const ov_core::ImuData &imu_1, const ov_core::ImuData &imu_2, double timestamp) {
  const double lambda = (timestamp - imu_1.timestamp) / (imu_2.timestamp - imu_1.timestamp);
  ov_core::ImuData data;
  data.timestamp = timestamp;
  for (int k = 0; k < 3; ++k) {
    data.am[k] = (1 - lambda) * imu_1.am[k] + lambda * imu_2.am[k];
    data.wm[k] = (1 - lambda) * imu_1.wm[k] + lambda * imu_2.wm[k];
  }
  return data;
}

std::vector<ov_core::ImuData> Propagator::select_imu_readings(const std::vector<ov_core::ImuData> &imu_data, double time0, double time1,
                                                              bool warn) {
  std::vector<ov_core::ImuData> prop_data;
  if (imu_data.empty()) {
    if (warn) PRINT_WARNING("Propagator::select_imu_readings(): No IMU measurements. IMU-CAMERA are likely messed up!!!\n");
    return prop_data;
  }
  for (size_t i = 0; i + 1 < imu_data.size(); i++) {
    const ov_core::ImuData &cur = imu_data[i], &nxt = imu_data[i + 1];
    if (nxt.timestamp > time0 && cur.timestamp < time0) {  // start of the period: split the straddling reading
      prop_data.push_back(interpolate_data(cur, nxt, time0));
      continue;
    }
    if (cur.timestamp >= time0 && nxt.timestamp <= time1) {  // fully inside
      prop_data.push_back(cur);
      continue;
    }
    if (nxt.timestamp > time1) {  // end of the period
      if (cur.timestamp > time1 && i == 0) {
        break;  // nothing before the start-up time: cannot propagate
      } else if (cur.timestamp > time1) {
        prop_data.push_back(interpolate_data(imu_data[i - 1], cur, time1));
      } else {
        prop_data.push_back(cur);
      }
      if (prop_data.back().timestamp != time1) prop_data.push_back(interpolate_data(cur, nxt, time1));
      break;
    }
  }
  if (prop_data.empty()) {
    if (warn) PRINT_WARNING("Propagator::select_imu_readings(): No IMU measurements to propagate with (0 of 2). IMU-CAMERA are likely messed up!!!\n");
    return prop_data;
  }
  for (size_t i = 0; i + 1 < prop_data.size(); i++) {
    if (std::abs(prop_data[i + 1].timestamp - prop_data[i].timestamp) < 1e-12) {
      if (warn) PRINT_WARNING("Propagator::select_imu_readings(): Zero DT between IMU reading %d and %d, removing it!\n", (int)i, (int)(i + 1));
      prop_data.erase(prop_data.begin() + i);
      i--;
    }
  }
  if (prop_data.size() < 2 && warn)
    PRINT_WARNING("Propagator::select_imu_readings(): No IMU measurements to propagate with (%d of 2). IMU-CAMERA are likely messed up!!!\n",
                  (int)prop_data.size());
  return prop_data;
}

void Propagator::propagate_and_clone(std::shared_ptr<State> state, double timestamp) {
  if (state->_timestamp == timestamp) {
    PRINT_ERROR("Propagator::propagate_and_clone(): Propagation called again at same timestep at last update timestep!!!!\n");
    std::exit(EXIT_FAILURE);
  }
  if (state->_timestamp > timestamp) {
    PRINT_ERROR("Propagator::propagate_and_clone(): Propagation called trying to propagate backwards in time!!!!\n");
    PRINT_ERROR("Propagator::propagate_and_clone(): desired propagation = %.4f\n", (timestamp - state->_timestamp));
    std::exit(EXIT_FAILURE);
  }
  if (!have_last_prop_time_offset) {
    last_prop_time_offset = state->_calib_dt_CAMtoIMU->value()(0);
    have_last_prop_time_offset = true;
  }
  const double t_off_new = state->_calib_dt_CAMtoIMU->value()(0);
  const double time0 = state->_timestamp + last_prop_time_offset;
  const double time1 = timestamp + t_off_new;
  std::vector<ov_core::ImuData> prop_data;
  {
    std::lock_guard<std::mutex> lck(imu_data_mtx);
    prop_data = Propagator::select_imu_readings(imu_data, time0, time1);
  }
  // Phi_summed = F_i Phi_summed ; Qd_summed = F_i Qd_summed F_i^T + Qd_i, symmetrised every interval (:92-102)
  double F[225], Qdi[225], T[225], T2[225];
  memset(_Phi, 0, sizeof(_Phi));
  memset(_Qs, 0, sizeof(_Qs));
  for (int i = 0; i < 15; ++i) _Phi[16 * i] = 1.0;
  if (prop_data.size() > 1) {
    for (size_t i = 0; i + 1 < prop_data.size(); i++) {
      predict_and_compute(state, prop_data[i], prop_data[i + 1], F, Qdi);
      mm15(F, _Phi, T);
      memcpy(_Phi, T, sizeof(T));
      mm15(F, _Qs, T);
      mmt15(T, F, T2);
      for (int k = 0; k < 225; ++k) T2[k] += Qdi[k];
      for (int c = 0; c < 15; ++c)
        for (int r = 0; r < 15; ++r) _Qs[15 * c + r] = 0.5 * (T2[15 * c + r] + T2[15 * r + c]);
    }
  }
  // last angular velocity for the time-offset Jacobian of the clone (:108-114)
  _last_w[0] = _last_w[1] = _last_w[2] = 0.0;
  if (prop_data.size() > 1) {
    for (int k = 0; k < 3; ++k) _last_w[k] = prop_data[prop_data.size() - 2].wm[k] - state->_imu->bias_g()[k];
  } else if (!prop_data.empty()) {
    for (int k = 0; k < 3; ++k) _last_w[k] = prop_data.back().wm[k] - state->_imu->bias_g()[k];
  }
  std::vector<std::shared_ptr<Type>> Phi_order;
  Phi_order.push_back(state->_imu);
  MatrixXd Phi_summed(15, 15), Qd_summed(15, 15);
  memcpy(Phi_summed.data(), _Phi, sizeof(_Phi));
  memcpy(Qd_summed.data(), _Qs, sizeof(_Qs));
  StateHelper::EKFPropagation(state, Phi_order, Phi_order, Phi_summed, Qd_summed);  // device: P strips only
  state->_timestamp = timestamp;
  last_prop_time_offset = t_off_new;
  StateHelper::augment_clone(state, _last_w);
}

void Propagator::predict_and_compute(std::shared_ptr<State> state, const ov_core::ImuData &data_minus, const ov_core::ImuData &data_plus,
                                     double F[225], double Qd[225]) {
  memset(F, 0, sizeof(double) * 225);
  memset(Qd, 0, sizeof(double) * 225);
  const double dt = data_plus.timestamp - data_minus.timestamp;
  double w_hat[3], a_hat[3], w_hat2[3], a_hat2[3];
  for (int k = 0; k < 3; ++k) {
    w_hat[k] = data_minus.wm[k] - state->_imu->bias_g()[k];
    a_hat[k] = data_minus.am[k] - state->_imu->bias_a()[k];
    w_hat2[k] = data_plus.wm[k] - state->_imu->bias_g()[k];
    a_hat2[k] = data_plus.am[k] - state->_imu->bias_a()[k];
  }
  double new_q[4], new_v[3], new_p[3];
  if (state->_options.use_rk4_integration) predict_mean_rk4(state, dt, w_hat, a_hat, w_hat2, a_hat2, new_q, new_v, new_p);
  else predict_mean_discrete(state, dt, w_hat, a_hat, w_hat2, a_hat2, new_q, new_v, new_p);
  const int th_id = state->_imu->q()->id() - state->_imu->id();
  const int p_id = state->_imu->p()->id() - state->_imu->id();
  const int v_id = state->_imu->v()->id() - state->_imu->id();
  const int bg_id = state->_imu->bg()->id() - state->_imu->id();
  const int ba_id = state->_imu->ba()->id() - state->_imu->id();
  double G[15 * 12];
  memset(G, 0, sizeof(G));
  const double mwdt[3] = {-w_hat[0] * dt, -w_hat[1] * dt, -w_hat[2] * dt};
  const M3 Jr = Jr_so3(mwdt), I3 = eye3();
  M3 Rth, RT;  // orientation block and the R^T that maps body to global in the v / p rows
  double av[3], ap[3];
  M3 Vth, Pth;
  if (state->_options.do_fej) {
    const M3 Rfej = from(state->_imu->Rot_fej());
    double Rn[9];
    quat_2_Rot(new_q, Rn);
    RT = tr(Rfej);
    Rth = mul(from(Rn), RT);
    for (int k = 0; k < 3; ++k) {
      av[k] = new_v[k] - state->_imu->vel_fej()[k] + _gravity[k] * dt;
      ap[k] = new_p[k] - state->_imu->pos_fej()[k] - state->_imu->vel_fej()[k] * dt + 0.5 * _gravity[k] * dt * dt;
    }
    Vth = mul(skew(av), RT);
    Pth = mul(skew(ap), RT);
    put(F, v_id, th_id, Vth, -1.0);
    put(F, p_id, th_id, Pth, -1.0);
  } else {
    RT = tr(from(state->_imu->Rot()));
    Rth = exp_so3(mwdt);
    for (int k = 0; k < 3; ++k) {
      av[k] = a_hat[k] * dt;
      ap[k] = a_hat[k] * dt * dt;
    }
    Vth = mul(RT, skew(av));
    Pth = mul(RT, skew(ap));
    put(F, v_id, th_id, Vth, -1.0);
    put(F, p_id, th_id, Pth, -0.5);
  }
  const M3 RJ = mul(Rth, Jr);
  put(F, th_id, th_id, Rth, 1.0);
  put(F, th_id, bg_id, RJ, -dt);
  put(F, bg_id, bg_id, I3, 1.0);
  put(F, v_id, v_id, I3, 1.0);
  put(F, v_id, ba_id, RT, -dt);
  put(F, ba_id, ba_id, I3, 1.0);
  put(F, p_id, v_id, I3, dt);
  put(F, p_id, ba_id, RT, -0.5 * dt * dt);
  put(F, p_id, p_id, I3, 1.0);
  put(G, th_id, 0, RJ, -dt);
  put(G, v_id, 3, RT, -dt);
  put(G, p_id, 3, RT, -0.5 * dt * dt);
  put(G, bg_id, 6, I3, 1.0);
  put(G, ba_id, 9, I3, 1.0);
  // Qd = G Qc G^T with Qc = diag(sw^2/dt, sa^2/dt, swb^2 dt, sab^2 dt) (x) I3, then symmetrised (:437-445)
  const double qc[4] = {_noises.sigma_w_2 / dt, _noises.sigma_a_2 / dt, _noises.sigma_wb_2 * dt, _noises.sigma_ab_2 * dt};
  double T[225];
  for (int j = 0; j < 15; ++j)
    for (int i = 0; i < 15; ++i) {
      double s = 0.0;
      for (int k = 0; k < 12; ++k) s += G[15 * k + i] * qc[k / 3] * G[15 * k + j];
      T[15 * j + i] = s;
    }
  for (int j = 0; j < 15; ++j)
    for (int i = 0; i < 15; ++i) Qd[15 * j + i] = 0.5 * (T[15 * j + i] + T[15 * i + j]);
  // value and first estimate both become the propagated mean (:447-453)
  VectorXd imu_x = state->_imu->value();
  for (int k = 0; k < 4; ++k) imu_x(k) = new_q[k];
  for (int k = 0; k < 3; ++k) {
    imu_x(4 + k) = new_p[k];
    imu_x(7 + k) = new_v[k];
  }
  state->_imu->set_value(imu_x);
  state->_imu->set_fej(imu_x);
}

void Propagator::predict_mean_discrete(std::shared_ptr<State> state, double dt, const double w1[3], const double a1[3], const double w2[3],
                                       const double a2[3], double new_q[4], double new_v[3], double new_p[3]) {
  double w[3], a[3];
  for (int k = 0; k < 3; ++k) {
    w[k] = state->_options.imu_avg ? .5 * (w1[k] + w2[k]) : w1[k];
    a[k] = state->_options.imu_avg ? .5 * (a1[k] + a2[k]) : a1[k];
  }
  const double w_norm = norm3(w);
  const double *q = state->_imu->quat();
  double Oq[4];
  Omega_times(w, q, Oq);
  // Trawny (101)/(103): closed-form zeroth-order quaternion integrator
  const double c0 = w_norm > 1e-20 ? std::cos(0.5 * w_norm * dt) : 1.0;
  const double c1 = w_norm > 1e-20 ? 1 / w_norm * std::sin(0.5 * w_norm * dt) : 0.5 * dt;
  for (int k = 0; k < 4; ++k) new_q[k] = c0 * q[k] + c1 * Oq[k];
  quatnorm(new_q);
  double Rta[3];
  tmul(from(state->_imu->Rot()), a, Rta);
  for (int k = 0; k < 3; ++k) {
    new_v[k] = state->_imu->vel()[k] + Rta[k] * dt - _gravity[k] * dt;
    new_p[k] = state->_imu->pos()[k] + state->_imu->vel()[k] * dt + 0.5 * Rta[k] * dt * dt - 0.5 * _gravity[k] * dt * dt;
  }
}

void Propagator::predict_mean_rk4(std::shared_ptr<State> state, double dt, const double w1[3], const double a1[3], const double w2[3],
                                  const double a2[3], double new_q[4], double new_v[3], double new_p[3]) {
  double w[3], a[3], w_alpha[3], a_jerk[3];
  for (int k = 0; k < 3; ++k) {
    w[k] = w1[k];
    a[k] = a1[k];
    w_alpha[k] = (w2[k] - w1[k]) / dt;
    a_jerk[k] = (a2[k] - a1[k]) / dt;
  }
  const double *q_0 = state->_imu->quat(), *p_0 = state->_imu->pos(), *v_0 = state->_imu->vel();
  // four stages on (dq, p, v): dq is the rotation since the start of the interval, re-normalised at every stage
  const double stage_frac[4] = {0.0, 0.5, 0.5, 1.0};
  const bool advance_imu[4] = {false, true, false, true};
  double kq[4][4], kp[4][3], kv[4][3];
  const double dq_0[4] = {0, 0, 0, 1};
  for (int s = 0; s < 4; ++s) {
    if (advance_imu[s])
      for (int k = 0; k < 3; ++k) {
        w[k] += 0.5 * w_alpha[k] * dt;
        a[k] += 0.5 * a_jerk[k] * dt;
      }
    double dq[4], v[3];
    for (int k = 0; k < 4; ++k) dq[k] = dq_0[k] + (s ? stage_frac[s] * kq[s - 1][k] : 0.0);
    if (s) quatnorm(dq);
    for (int k = 0; k < 3; ++k) v[k] = v_0[k] + (s ? stage_frac[s] * kv[s - 1][k] : 0.0);
    double q_dot[4], qs[4], Rs[9], Rta[3];
    Omega_times(w, dq, q_dot);
    quat_multiply(dq, q_0, qs);
    quat_2_Rot(qs, Rs);
    tmul(from(Rs), a, Rta);
    for (int k = 0; k < 4; ++k) kq[s][k] = 0.5 * q_dot[k] * dt;
    for (int k = 0; k < 3; ++k) {
      kp[s][k] = v[k] * dt;
      kv[s][k] = (Rta[k] - _gravity[k]) * dt;
    }
  }
  double dq[4];
  for (int k = 0; k < 4; ++k)
    dq[k] = dq_0[k] + (1.0 / 6.0) * kq[0][k] + (1.0 / 3.0) * kq[1][k] + (1.0 / 3.0) * kq[2][k] + (1.0 / 6.0) * kq[3][k];
  quatnorm(dq);
  quat_multiply(dq, q_0, new_q);
  for (int k = 0; k < 3; ++k) {
    new_p[k] = p_0[k] + (1.0 / 6.0) * kp[0][k] + (1.0 / 3.0) * kp[1][k] + (1.0 / 3.0) * kp[2][k] + (1.0 / 6.0) * kp[3][k];
    new_v[k] = v_0[k] + (1.0 / 6.0) * kv[0][k] + (1.0 / 3.0) * kv[1][k] + (1.0 / 3.0) * kv[2][k] + (1.0 / 6.0) * kv[3][k];
  }
}

}  // namespace ov_plane
