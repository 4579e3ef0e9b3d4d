// Host mirrors of UpdaterHelper (dense Jacobians for the small SLAM systems), UpdaterSLAM and UpdaterPlane.
// The covariance work goes to the device through StateHelper (EKFUpdate / initialize / get_marginal_covariance) or the
// batched C-ABI calls; what stays here is what is host scalar code in the reference as well.
#include <chrono>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>

#include "ov_plane_host.h"

using namespace ov_type;

namespace ov_plane {

static void gpu_check2(int rc, const char *what) {
  if (rc == 0) return;
  fprintf(stderr, "ov_plane(gpu): %s failed: %s\n", what, ovp_error_string(rc));
  std::exit(EXIT_FAILURE);
}

static inline void m3v(const double *A, const double *v, double *o) {
  for (int i = 0; i < 3; ++i) o[i] = A[3 * i] * v[0] + A[3 * i + 1] * v[1] + A[3 * i + 2] * v[2];
}
static inline void m3m(const double *A, const double *B, double *C) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
static inline void skew3(const double *w, double *S) {
  S[0] = 0;
  S[1] = -w[2];
  S[2] = w[1];
  S[3] = w[2];
  S[4] = 0;
  S[5] = -w[0];
  S[6] = -w[1];
  S[7] = w[0];
  S[8] = 0;
}

// ---- update/UpdaterHelper.cpp:35-193 -------------------------------------------------------------
static void inv_depth_jac(const double p[3], MatrixXd &J) {  // d p / d (theta, phi, rho), :50-72
  const double rho = 1 / std::sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
  const double phi = std::acos(rho * p[2]), th = std::atan2(p[1], p[0]);
  const double st = std::sin(th), ct = std::cos(th), sp = std::sin(phi), cp = std::cos(phi);
  J = MatrixXd::Zero(3, 3);
  J(0, 0) = -(1.0 / rho) * st * sp;
  J(0, 1) = (1.0 / rho) * ct * cp;
  J(0, 2) = -(1.0 / (rho * rho)) * ct * sp;
  J(1, 0) = (1.0 / rho) * ct * sp;
  J(1, 1) = (1.0 / rho) * st * cp;
  J(1, 2) = -(1.0 / (rho * rho)) * st * sp;
  J(2, 1) = -(1.0 / rho) * sp;
  J(2, 2) = -(1.0 / (rho * rho)) * cp;
}

void UpdaterHelper::get_feature_jacobian_representation(std::shared_ptr<State> state, UpdaterHelperFeature &feature, MatrixXd &H_f,
                                                        std::vector<MatrixXd> &H_x, std::vector<std::shared_ptr<Type>> &x_order) {
  typedef LandmarkRepresentation LR;
  if (feature.feat_representation == LR::GLOBAL_3D) {  // :39-43
    H_f = MatrixXd::Zero(3, 3);
    for (int k = 0; k < 3; ++k) H_f(k, k) = 1.0;
    return;
  }
  if (feature.feat_representation == LR::GLOBAL_FULL_INVERSE_DEPTH) {  // :46-76
    inv_depth_jac(state->_options.do_fej ? feature.p_FinG_fej : feature.p_FinG, H_f);
    return;
  }
  assert(feature.anchor_cam_id != -1);  // :83
  auto calib = state->_calib_IMUtoCAM.at(feature.anchor_cam_id);
  auto anchor = state->_clones_IMU.at(feature.anchor_clone_timestamp);
  const double *R_ItoC = calib->Rot(), *p_IinC = calib->pos();
  const double *R_GtoI = anchor->Rot(), *p_IinG = anchor->pos();
  double p_FinA[3] = {feature.p_FinA[0], feature.p_FinA[1], feature.p_FinA[2]};
  if (state->_options.do_fej) {  // :91-98
    double q[3] = {p_FinA[0] - p_IinC[0], p_FinA[1] - p_IinC[1], p_FinA[2] - p_IinC[2]}, t[3], best[3];
    for (int i = 0; i < 3; ++i) t[i] = R_ItoC[i] * q[0] + R_ItoC[3 + i] * q[1] + R_ItoC[6 + i] * q[2];
    for (int i = 0; i < 3; ++i) best[i] = R_GtoI[i] * t[0] + R_GtoI[3 + i] * t[1] + R_GtoI[6 + i] * t[2] + p_IinG[i];
    R_GtoI = anchor->Rot_fej();
    p_IinG = anchor->pos_fej();
    const double d[3] = {best[0] - p_IinG[0], best[1] - p_IinG[1], best[2] - p_IinG[2]};
    m3v(R_GtoI, d, t);
    m3v(R_ItoC, t, p_FinA);
    for (int k = 0; k < 3; ++k) p_FinA[k] += p_IinC[k];
  }
  MatrixXd R_CtoG(3, 3);  // R_GtoI^T R_ItoC^T
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double a = 0;
      for (int k = 0; k < 3; ++k) a += R_GtoI[3 * k + i] * R_ItoC[3 * j + k];
      R_CtoG(i, j) = a;
    }
  const double q[3] = {p_FinA[0] - p_IinC[0], p_FinA[1] - p_IinC[1], p_FinA[2] - p_IinC[2]};
  {  // :102-109
    double v[3], S[9];
    for (int i = 0; i < 3; ++i) v[i] = R_ItoC[i] * q[0] + R_ItoC[3 + i] * q[1] + R_ItoC[6 + i] * q[2];
    skew3(v, S);
    MatrixXd H_anc = MatrixXd::Zero(3, 6);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        double a = 0;
        for (int k = 0; k < 3; ++k) a += R_GtoI[3 * k + i] * S[3 * k + j];
        H_anc(i, j) = -a;
        H_anc(i, 3 + j) = (i == j) ? 1.0 : 0.0;
      }
    x_order.push_back(anchor);
    H_x.push_back(H_anc);
  }
  if (state->_options.do_calib_camera_pose) {  // :112-119
    double S[9];
    skew3(q, S);
    MatrixXd H_calib = MatrixXd::Zero(3, 6);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        double a = 0;
        for (int k = 0; k < 3; ++k) a += R_CtoG(i, k) * S[3 * k + j];
        H_calib(i, j) = -a;
        H_calib(i, 3 + j) = -R_CtoG(i, j);
      }
    x_order.push_back(calib);
    H_x.push_back(H_calib);
  }
  MatrixXd J;
  if (feature.feat_representation == LR::ANCHORED_3D) {  // :122-125
    H_f = R_CtoG;
    return;
  }
  if (feature.feat_representation == LR::ANCHORED_FULL_INVERSE_DEPTH) {  // :128-152
    inv_depth_jac(p_FinA, J);
  } else if (feature.feat_representation == LR::ANCHORED_MSCKF_INVERSE_DEPTH) {  // :155-173
    const double rho = 1 / p_FinA[2], al = p_FinA[0] / p_FinA[2], be = p_FinA[1] / p_FinA[2];
    J = MatrixXd::Zero(3, 3);
    J(0, 0) = 1.0 / rho;
    J(0, 2) = -(1.0 / (rho * rho)) * al;
    J(1, 1) = 1.0 / rho;
    J(1, 2) = -(1.0 / (rho * rho)) * be;
    J(2, 2) = -(1.0 / (rho * rho));
  } else if (feature.feat_representation == LR::ANCHORED_INVERSE_DEPTH_SINGLE) {  // :176-187
    const double rho = 1.0 / p_FinA[2];
    H_f = MatrixXd::Zero(3, 1);
    for (int i = 0; i < 3; ++i) {
      double a = 0;
      for (int k = 0; k < 3; ++k) a += R_CtoG(i, k) * (-(1.0 / (rho * rho)) * (rho * p_FinA[k]));
      H_f(i, 0) = a;
    }
    return;
  } else {
    assert(false);  // :190
  }
  H_f = MatrixXd::Zero(3, 3);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double a = 0;
      for (int k = 0; k < 3; ++k) a += R_CtoG(i, k) * J(k, j);
      H_f(i, j) = a;
    }
}

// ---- update/UpdaterHelper.cpp:195-513 (every landmark representation, radtan / equidistant, any number of cameras) ---
void UpdaterHelper::get_feature_jacobian_full(std::shared_ptr<State> state, UpdaterHelperFeature &feature, double sigma_px,
                                              double sigma_c, MatrixXd &H_f, MatrixXd &H_x, VectorXd &res,
                                              std::vector<std::shared_ptr<Type>> &x_order) {
  const int total_meas = (int)feature.timestamps.size();
  x_order.clear();
  int total_hx = 0;
  std::vector<std::pair<std::shared_ptr<Type>, int>> map_hx;
  auto find_col = [&](const std::shared_ptr<Type> &v) {
    for (auto &p : map_hx)
      if (p.first == v) return p.second;
    return -1;
  };
  // :205-228 every camera that measured the feature brings its extrinsics / intrinsics (when they are estimated), in camera order
  auto cam_of = [&](int m) { return (size_t)m < feature.cam_ids.size() ? feature.cam_ids[m] : 0; };
  {
    std::vector<int> cams;
    for (int m = 0; m < total_meas; m++) cams.push_back(cam_of(m));
    if (cams.empty()) cams.push_back(0);
    std::sort(cams.begin(), cams.end());
    cams.erase(std::unique(cams.begin(), cams.end()), cams.end());
    for (int cam : cams) {
      std::shared_ptr<PoseJPL> calibration = state->_calib_IMUtoCAM.at((size_t)cam);
      std::shared_ptr<Vec> distortion = state->_cam_intrinsics.at((size_t)cam);
      if (state->_options.do_calib_camera_pose) {  // :216-220
        map_hx.push_back({calibration, total_hx});
        x_order.push_back(calibration);
        total_hx += calibration->size();
      }
      if (state->_options.do_calib_camera_intrinsics) {  // :223-227
        map_hx.push_back({distortion, total_hx});
        x_order.push_back(distortion);
        total_hx += distortion->size();
      }
    }
  }
  for (int m = 0; m < total_meas; m++) {  // :230-239
    std::shared_ptr<PoseJPL> clone_Ci = state->_clones_IMU.at(feature.timestamps[m]);
    if (find_col(clone_Ci) < 0) {
      map_hx.push_back({clone_Ci, total_hx});
      x_order.push_back(clone_Ci);
      total_hx += clone_Ci->size();
    }
  }
  const bool relative = LandmarkRepresentation::is_relative_representation(feature.feat_representation);
  if (relative) {  // :241-263 the anchor clone (and its extrinsics) are part of the system
    std::shared_ptr<PoseJPL> clone_anchor = state->_clones_IMU.at(feature.anchor_clone_timestamp);
    if (find_col(clone_anchor) < 0) {
      map_hx.push_back({clone_anchor, total_hx});
      x_order.push_back(clone_anchor);
      total_hx += clone_anchor->size();
    }
  }
  const bool plane_in_state = (state->_features_PLANE.find(feature.planeid) != state->_features_PLANE.end());  // :269
  if (feature.planeid != 0 && plane_in_state) {
    std::shared_ptr<Vec> planecp = state->_features_PLANE.at(feature.planeid);
    if (find_col(planecp) < 0) {
      map_hx.push_back({planecp, total_hx});
      x_order.push_back(planecp);
      total_hx += planecp->size();
    }
  }
  // :283-302 position in the global frame; for the anchored representations the "best" estimate serves as FEJ value too
  double p_FinG_buf[3];
  if (relative) {
    auto calib_a = state->_calib_IMUtoCAM.at(feature.anchor_cam_id);
    auto anchor = state->_clones_IMU.at(feature.anchor_clone_timestamp);
    const double *Rc = calib_a->Rot(), *pc = calib_a->pos(), *Ra = anchor->Rot(), *pa = anchor->pos();
    const double q[3] = {feature.p_FinA[0] - pc[0], feature.p_FinA[1] - pc[1], feature.p_FinA[2] - pc[2]};
    double t[3];
    for (int i = 0; i < 3; ++i) t[i] = Rc[i] * q[0] + Rc[3 + i] * q[1] + Rc[6 + i] * q[2];
    for (int i = 0; i < 3; ++i) p_FinG_buf[i] = Ra[i] * t[0] + Ra[3 + i] * t[1] + Ra[6 + i] * t[2] + pa[i];
  }
  const double *p_FinG = relative ? p_FinG_buf : feature.p_FinG;
  const double *p_FinG_fej = relative ? p_FinG_buf : feature.p_FinG_fej;
  int c = 0;
  const int nlam = (feature.feat_representation != LandmarkRepresentation::ANCHORED_INVERSE_DEPTH_SINGLE) ? 3 : 1;
  int jacobsize = nlam + ((feature.planeid != 0 && !plane_in_state) ? 3 : 0);  // :310-311
  // :323-327 representation Jacobians, once per feature
  MatrixXd dpfg_dlambda;
  std::vector<MatrixXd> dpfg_dx;
  std::vector<std::shared_ptr<Type>> dpfg_dx_order;
  get_feature_jacobian_representation(state, feature, dpfg_dlambda, dpfg_dx, dpfg_dx_order);
  int meassize = (feature.planeid != 0) ? (3 * total_meas) : (2 * total_meas);
  if (total_meas == 0 && feature.planeid != 0) meassize = 1;
  res = VectorXd::Zero(meassize, 1);
  H_f = MatrixXd::Zero(meassize, jacobsize);
  H_x = MatrixXd::Zero(meassize, total_hx);
  const double white_px = 1.0 / sigma_px;
  for (int m = 0; m < total_meas; m++) {
    // the camera that took this measurement (:335-344 loops over the cameras, then over each one's measurements)
    const size_t cam = (size_t)cam_of(m);
    std::shared_ptr<PoseJPL> calibration = state->_calib_IMUtoCAM.at(cam);
    std::shared_ptr<Vec> distortion = state->_cam_intrinsics.at(cam);
    const double *R_ItoC = calibration->Rot();
    const double *p_IinC = calibration->pos();
    const double *v = distortion->value().data();
    std::shared_ptr<PoseJPL> clone_Ii = state->_clones_IMU.at(feature.timestamps[m]);
    const double *R_GtoIi = clone_Ii->Rot();
    const double *p_IiinG = clone_Ii->pos();
    double d[3] = {p_FinG[0] - p_IiinG[0], p_FinG[1] - p_IiinG[1], p_FinG[2] - p_IiinG[2]};
    double p_FinIi[3], p_FinCi[3];
    m3v(R_GtoIi, d, p_FinIi);
    m3v(R_ItoC, p_FinIi, p_FinCi);
    for (int k = 0; k < 3; ++k) p_FinCi[k] += p_IinC[k];
    const double x = p_FinCi[0] / p_FinCi[2], y = p_FinCi[1] / p_FinCi[2];
    // ext CamRadtan::distort_d / CamEqui::distort_d (:365)
    const bool fisheye = state->_cam_fisheye.count(cam) && state->_cam_fisheye.at(cam);
    const double r2 = x * x + y * y, r4 = r2 * r2, g = 1 + v[4] * r2 + v[5] * r4;
    double x1, y1;
    double rr = 0, th = 0, th_d = 0, inv_r = 1, cdist = 1;
    if (fisheye) {
      rr = std::sqrt(r2);
      th = std::atan(rr);
      th_d = th + v[4] * std::pow(th, 3) + v[5] * std::pow(th, 5) + v[6] * std::pow(th, 7) + v[7] * std::pow(th, 9);
      inv_r = (rr > 1e-8) ? 1.0 / rr : 1.0;
      cdist = (rr > 1e-8) ? th_d * inv_r : 1.0;
      x1 = x * cdist;
      y1 = y * cdist;
    } else {
      x1 = x * g + 2 * v[6] * x * y + v[7] * (r2 + 2 * x * x);
      y1 = y * g + v[6] * (r2 + 2 * y * y) + 2 * v[7] * x * y;
    }
    res(c) = white_px * ((double)feature.uvs[2 * m] - (v[0] * x1 + v[2]));
    res(c + 1) = white_px * ((double)feature.uvs[2 * m + 1] - (v[1] * y1 + v[3]));
    if (state->_options.do_fej) {  // :376-385
      R_GtoIi = clone_Ii->Rot_fej();
      p_IiinG = clone_Ii->pos_fej();
      for (int k = 0; k < 3; ++k) d[k] = p_FinG_fej[k] - p_IiinG[k];
      m3v(R_GtoIi, d, p_FinIi);
      m3v(R_ItoC, p_FinIi, p_FinCi);
      for (int k = 0; k < 3; ++k) p_FinCi[k] += p_IinC[k];
    }
    // ext CamRadtan::compute_distort_jacobian at the non-FEJ uv_norm (:389)
    const double fx = v[0], fy = v[1], k1 = v[4], k2 = v[5], p1 = v[6], p2 = v[7];
    double dz_dzn[4], dz_dzeta[16];
    dz_dzn[0] = fx * (g + 2 * k1 * x * x + 4 * k2 * x * x * r2 + 2 * p1 * y + 6 * p2 * x);
    dz_dzn[1] = fx * (2 * k1 * x * y + 4 * k2 * x * y * r2 + 2 * p1 * x + 2 * p2 * y);
    dz_dzn[2] = fy * (2 * k1 * x * y + 4 * k2 * x * y * r2 + 2 * p1 * x + 2 * p2 * y);
    dz_dzn[3] = fy * (g + 2 * k1 * y * y + 4 * k2 * y * y * r2 + 6 * p1 * y + 2 * p2 * x);
    memset(dz_dzeta, 0, sizeof(dz_dzeta));
    dz_dzeta[0] = x1;
    dz_dzeta[2] = 1;
    dz_dzeta[4] = fx * x * r2;
    dz_dzeta[5] = fx * x * r4;
    dz_dzeta[6] = 2 * fx * x * y;
    dz_dzeta[7] = fx * (r2 + 2 * x * x);
    dz_dzeta[9] = y1;
    dz_dzeta[11] = 1;
    dz_dzeta[12] = fy * y * r2;
    dz_dzeta[13] = fy * y * r4;
    dz_dzeta[14] = fy * (r2 + 2 * y * y);
    dz_dzeta[15] = 2 * fy * x * y;
    if (fisheye) {  // ext CamEqui::compute_distort_jacobian
      const double dthd_dth = 1 + 3 * v[4] * std::pow(th, 2) + 5 * v[5] * std::pow(th, 4) + 7 * v[6] * std::pow(th, 6) + 9 * v[7] * std::pow(th, 8);
      const double dth_dr = 1 / (rr * rr + 1);
      const double a0 = -x * th_d * inv_r * inv_r + x * inv_r * dthd_dth * dth_dr;
      const double a1 = -y * th_d * inv_r * inv_r + y * inv_r * dthd_dth * dth_dr;
      dz_dzn[0] = fx * (th_d * inv_r + a0 * x * inv_r);
      dz_dzn[1] = fx * (a0 * y * inv_r);
      dz_dzn[2] = fy * (a1 * x * inv_r);
      dz_dzn[3] = fy * (th_d * inv_r + a1 * y * inv_r);
      memset(dz_dzeta, 0, sizeof(dz_dzeta));
      dz_dzeta[0] = x1;
      dz_dzeta[2] = 1;
      dz_dzeta[9] = y1;
      dz_dzeta[11] = 1;
      for (int k = 0; k < 4; ++k) {
        const double pw = std::pow(th, 3 + 2 * k);
        dz_dzeta[4 + k] = fx * x * inv_r * pw;
        dz_dzeta[12 + k] = fy * y * inv_r * pw;
      }
    }
    const double z = p_FinCi[2];
    const double dzn_dpfc[6] = {1 / z, 0, -p_FinCi[0] / (z * z), 0, 1 / z, -p_FinCi[1] / (z * z)};
    double dpfc_dpfg[9], sk[9], Rsk[9];
    m3m(R_ItoC, R_GtoIi, dpfc_dpfg);
    skew3(p_FinIi, sk);
    m3m(R_ItoC, sk, Rsk);
    double dz_dpfc[6], dz_dpfg[6];
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 3; ++j) dz_dpfc[3 * i + j] = dz_dzn[2 * i] * dzn_dpfc[j] + dz_dzn[2 * i + 1] * dzn_dpfc[3 + j];
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 3; ++j)
        dz_dpfg[3 * i + j] = dz_dpfc[3 * i] * dpfc_dpfg[j] + dz_dpfc[3 * i + 1] * dpfc_dpfg[3 + j] + dz_dpfc[3 * i + 2] * dpfc_dpfg[6 + j];
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < nlam; ++j)  // :411
        H_f(c + i, j) = white_px * (dz_dpfg[3 * i] * dpfg_dlambda(0, j) + dz_dpfg[3 * i + 1] * dpfg_dlambda(1, j) + dz_dpfg[3 * i + 2] * dpfg_dlambda(2, j));
    const int cc = find_col(clone_Ii);
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 3; ++j) {  // :414
        H_x(c + i, cc + j) = white_px * (dz_dpfc[3 * i] * Rsk[j] + dz_dpfc[3 * i + 1] * Rsk[3 + j] + dz_dpfc[3 * i + 2] * Rsk[6 + j]);
        H_x(c + i, cc + 3 + j) = -white_px * dz_dpfg[3 * i + j];
      }
    for (size_t e = 0; e < dpfg_dx_order.size(); ++e) {  // :419-421 (+= : this may be the anchoring pose itself)
      const int ce = find_col(dpfg_dx_order[e]);
      for (int i = 0; i < 2; ++i)
        for (int j = 0; j < dpfg_dx_order[e]->size(); ++j)
          H_x(c + i, ce + j) += white_px * (dz_dpfg[3 * i] * dpfg_dx[e](0, j) + dz_dpfg[3 * i + 1] * dpfg_dx[e](1, j) + dz_dpfg[3 * i + 2] * dpfg_dx[e](2, j));
    }
    if (state->_options.do_calib_camera_pose) {  // :426-435
      const double w[3] = {p_FinCi[0] - p_IinC[0], p_FinCi[1] - p_IinC[1], p_FinCi[2] - p_IinC[2]};
      double skc[9];
      skew3(w, skc);
      const int ccal = find_col(calibration);
      for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 3; ++j) {
          H_x(c + i, ccal + j) += white_px * (dz_dpfc[3 * i] * skc[j] + dz_dpfc[3 * i + 1] * skc[3 + j] + dz_dpfc[3 * i + 2] * skc[6 + j]);
          H_x(c + i, ccal + 3 + j) += white_px * dz_dpfc[3 * i + j];
        }
    }
    if (state->_options.do_calib_camera_intrinsics) {  // :438-440
      const int cin = find_col(distortion);
      for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 8; ++j) H_x(c + i, cin + j) = white_px * dz_dzeta[8 * i + j];
    }
    c += 2;
  }
  if (feature.planeid != 0) {  // :448-512
    const double white_c = 1.0 / sigma_c;
    const int reps = (total_meas == 0) ? 1 : total_meas;
    for (int rep = 0; rep < reps; ++rep) {
      const double *cp = feature.cp_FinG;
      double dd = std::sqrt(cp[0] * cp[0] + cp[1] * cp[1] + cp[2] * cp[2]);
      double n[3] = {cp[0] / dd, cp[1] / dd, cp[2] / dd};
      res(c) = white_c * (0.0 - (n[0] * p_FinG[0] + n[1] * p_FinG[1] + n[2] * p_FinG[2] - dd));
      const double *lp = p_FinG;
      if (state->_options.do_fej) {
        lp = p_FinG_fej;
        cp = feature.cp_FinG_fej;
        dd = std::sqrt(cp[0] * cp[0] + cp[1] * cp[1] + cp[2] * cp[2]);
        for (int k = 0; k < 3; ++k) n[k] = cp[k] / dd;
      }
      const double ndp = n[0] * lp[0] + n[1] * lp[1] + n[2] * lp[2];
      for (int j = 0; j < 3; ++j) {
        const double hcp = white_c * 1.0 / dd * (lp[j] - ndp * n[j] - dd * n[j]);
        if (plane_in_state) H_x(c, find_col(state->_features_PLANE.at(feature.planeid)) + j) = hcp;
        else H_f(c, jacobsize - 3 + j) = hcp;
        H_f(c, j) = white_c * n[j];
      }
      c += 1;
    }
  }
}

static void givens(double p, double q, double &c, double &s) {
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
static void rot2(MatrixXd &A, int r0, int c0, double c, double s) {
  for (int j = c0; j < A.cols(); ++j) {
    const double x = A(r0, j), y = A(r0 + 1, j);
    A(r0, j) = c * x - s * y;
    A(r0 + 1, j) = s * x + c * y;
  }
}

// ---- update/UpdaterHelper.cpp:515-546 ------------------------------------------------------------
void UpdaterHelper::nullspace_project_inplace(MatrixXd &H_f, MatrixXd &H_x, VectorXd &res) {
  assert(H_f.rows() >= H_f.cols());
  for (int n = 0; n < H_f.cols(); ++n)
    for (int m = H_f.rows() - 1; m > n; m--) {
      double c, s;
      givens(H_f(m - 1, n), H_f(m, n), c, s);
      rot2(H_f, m - 1, n, c, s);
      rot2(H_x, m - 1, 0, c, s);
      rot2(res, m - 1, 0, c, s);
    }
  H_x = H_x.block(H_f.cols(), 0, H_x.rows() - H_f.cols(), H_x.cols());
  res = res.block(H_f.cols(), 0, res.rows() - H_f.cols(), res.cols());
}

// ---- update/UpdaterHelper.cpp:548-579 ------------------------------------------------------------
void UpdaterHelper::measurement_compress_inplace(MatrixXd &H_x, VectorXd &res) {
  if (H_x.rows() <= H_x.cols()) return;
  for (int n = 0; n < H_x.cols(); n++)
    for (int m = H_x.rows() - 1; m > n; m--) {
      double c, s;
      givens(H_x(m - 1, n), H_x(m, n), c, s);
      rot2(H_x, m - 1, n, c, s);
      rot2(res, m - 1, 0, c, s);
    }
  const int r = std::min(H_x.rows(), H_x.cols());
  H_x = H_x.block(0, 0, r, H_x.cols());
  res = res.block(0, 0, r, 1);
}

// Pose tables of the clone window + camera calibration -> device (ovp_state_upload); clone_slot: timestamp -> clone slot
void UpdaterSLAM::upload_state_tables(std::shared_ptr<State> state, std::map<double, int> &clone_slot,
                                      std::vector<std::shared_ptr<PoseJPL>> &clones) {
  clone_slot.clear();
  clones.clear();
  for (const auto &c : state->_clones_IMU) {
    clone_slot[c.first] = (int)clones.size();
    clones.push_back(c.second);
  }
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
  ovp_state_tables st;
  st.n_state = ovp_cov_size(state->_gpu);
  st.n_clones = C;
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
  gpu_check2(ovp_state_upload(state->_gpu, &st), "ovp_state_upload");  // (the tables are staged inside the call)
}

// ---- update/UpdaterSLAM.cpp ----------------------------------------------------------------------
UpdaterSLAM::UpdaterSLAM(UpdaterOptions &options_slam, UpdaterOptions &options_aruco, ov_core::FeatureInitializerOptions &fio)
    : _options_slam(options_slam), _options_aruco(options_aruco), _featinit(fio) {
  _options_slam.sigma_pix_sq = std::pow(_options_slam.sigma_pix, 2);
  _options_aruco.sigma_pix_sq = std::pow(_options_aruco.sigma_pix, 2);
}

static void clean_old_measurements(ov_core::Feature &ft, const std::map<double, std::shared_ptr<PoseJPL>> &clones) {
  std::vector<float> uv2, uvn2;
  std::vector<double> ts2;
  std::vector<int> cam2;
  const bool has_norm = ft.uvs_norm.size() == ft.uvs.size();
  for (size_t k = 0; k < ft.timestamps.size(); ++k)
    if (clones.count(ft.timestamps[k])) {
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
}

// update/UpdaterSLAM.cpp:376-682.  The rows of a GLOBAL_3D landmark, every chi2 test (against the RESIDENT covariance), the
// no-plane fallback, the stacking and StateHelper::EKFUpdate are one device call (ovp_slam_update, csrc/k_slam.hip); what stays
// here is the bookkeeping of the feature vector and the representation Jacobians of landmarks that are not GLOBAL_3D
// (update/UpdaterHelper.cpp:35-193), whose dense blocks ride in the same call.
void UpdaterSLAM::update(std::shared_ptr<State> state, std::vector<std::shared_ptr<ov_core::Feature>> &feature_vec,
                         const std::map<size_t, size_t> &feat2plane) {
  if (feature_vec.empty()) return;
  typedef LandmarkRepresentation LR;
  // :391-419
  auto it0 = feature_vec.begin();
  while (it0 != feature_vec.end()) {
    clean_old_measurements(**it0, state->_clones_IMU);
    const int ct_meas = (int)(*it0)->timestamps.size();
    std::shared_ptr<Landmark> landmark = state->_features_SLAM.at((*it0)->featid);
    const int required_meas = (landmark->_feat_representation == LR::ANCHORED_INVERSE_DEPTH_SINGLE) ? 2 : 1;  // :409-410
    if (ct_meas < 1) {
      (*it0)->to_delete = true;
      it0 = feature_vec.erase(it0);
    } else if (ct_meas < required_meas) {
      it0 = feature_vec.erase(it0);  // :416-418 (not flagged)
    } else {
      it0++;
    }
  }
  if (feature_vec.empty()) return;
  std::map<double, int> clone_slot;
  std::vector<std::shared_ptr<PoseJPL>> clones;
  UpdaterSLAM::upload_state_tables(state, clone_slot, clones);
  const int L = (int)feature_vec.size();
  int M = 1;
  bool other_camera = false;
  for (auto &f : feature_vec) {
    M = std::max(M, (int)f->timestamps.size());
    other_camera = other_camera || !f->only_camera0();
  }
  // a track longer than one wavefront's rows, or measurements of a camera other than camera 0 (the device batch carries one
  // calibration block per row): the dense form has neither limit
  if (M > OVP_MAX_MEAS || other_camera || _force_dense) {
    update_dense(state, feature_vec, feat2plane);
    return;
  }
  std::vector<float> uv((size_t)L * M * 2, 0.f);
  std::vector<int> cidx((size_t)L * M, -1), nm(L), lmid(L), psid(L, -1), pre_rows(L, 0), pre_cols(L, 0), pre_ids;
  std::vector<double> pv((size_t)L * 3, 0.0), pf((size_t)L * 3, 0.0), cpv((size_t)L * 3, 0.0), cpf((size_t)L * 3, 0.0), pre_H;
  std::vector<size_t> planeid(L, 0);
  bool any_pre = false, any_plane = false;
  const double sigma_c = state->_options.sigma_constraint;
  for (int l = 0; l < L; ++l) {
    ov_core::Feature &ft = *feature_vec[l];
    std::shared_ptr<Landmark> landmark = state->_features_SLAM.at(ft.featid);
    nm[l] = (int)ft.timestamps.size();
    lmid[l] = landmark->id();
    for (int k = 0; k < nm[l]; ++k) {
      cidx[(size_t)l * M + k] = clone_slot.at(ft.timestamps[k]);
      uv[((size_t)l * M + k) * 2] = ft.uvs[2 * k];
      uv[((size_t)l * M + k) * 2 + 1] = ft.uvs[2 * k + 1];
    }
    // :465-475
    if (state->_options.use_plane_constraint && state->_options.use_plane_constraint_slamu && feat2plane.find(ft.featid) != feat2plane.end() &&
        state->_features_PLANE.find(feat2plane.at(ft.featid)) != state->_features_PLANE.end()) {
      if (state->_features_SLAM_to_PLANE.find(ft.featid) == state->_features_SLAM_to_PLANE.end() ||
          state->_features_SLAM_to_PLANE.at(ft.featid) != 0) {
        planeid[l] = feat2plane.at(ft.featid);
        auto pl = state->_features_PLANE.at(planeid[l]);
        psid[l] = pl->id();
        for (int k = 0; k < 3; ++k) {
          cpv[3 * l + k] = pl->value()(k);
          cpf[3 * l + k] = pl->fej()(k);
        }
        any_plane = true;
      }
    }
    if (landmark->_feat_representation == LR::GLOBAL_3D) {
      landmark->get_xyz(false, &pv[3 * l]);
      landmark->get_xyz(true, &pf[3 * l]);
      continue;
    }
    // :478-522 the landmark in its own representation: dense block [H_x | H_f] from the host
    if (planeid[l] != 0) {  // update/UpdaterHelper.cpp:455-456 asserts GLOBAL_3D for the point-on-plane rows
      fprintf(stderr, "UpdaterSLAM::update() - point-on-plane rows need a GLOBAL_3D landmark\n");
      std::exit(EXIT_FAILURE);
    }
    UpdaterHelper::UpdaterHelperFeature feat;
    feat.featid = ft.featid;
    feat.uvs = ft.uvs;
    feat.timestamps = ft.timestamps;
    const bool single = landmark->_feat_representation == LR::ANCHORED_INVERSE_DEPTH_SINGLE;
    feat.feat_representation = single ? LR::ANCHORED_MSCKF_INVERSE_DEPTH : landmark->_feat_representation;  // :478-481
    if (LR::is_relative_representation(feat.feat_representation)) {
      feat.anchor_cam_id = landmark->_anchor_cam_id;
      feat.anchor_clone_timestamp = landmark->_anchor_clone_timestamp;
      landmark->get_xyz(false, feat.p_FinA);
      landmark->get_xyz(true, feat.p_FinA_fej);
    } else {
      landmark->get_xyz(false, feat.p_FinG);
      landmark->get_xyz(true, feat.p_FinG_fej);
    }
    MatrixXd H_f, H_x;
    VectorXd res;
    std::vector<std::shared_ptr<Type>> Hx_order;
    UpdaterHelper::get_feature_jacobian_full(state, feat, _options_slam.sigma_pix, sigma_c, H_f, H_x, res, Hx_order);
    MatrixXd H_xf;
    if (single) {  // :499-515 the depth column joins the state side, the bearing is projected out
      H_xf = MatrixXd(H_x.rows(), H_x.cols() + 1);
      for (int j = 0; j < H_x.cols(); ++j)
        for (int i = 0; i < H_x.rows(); ++i) H_xf(i, j) = H_x(i, j);
      for (int i = 0; i < H_x.rows(); ++i) H_xf(i, H_x.cols()) = H_f(i, H_f.cols() - 1);
      MatrixXd H_b = H_f.block(0, 0, H_f.rows(), H_f.cols() - 1);
      UpdaterHelper::nullspace_project_inplace(H_b, H_xf, res);
    } else {  // :517-522
      H_xf = MatrixXd(H_x.rows(), H_x.cols() + H_f.cols());
      for (int j = 0; j < H_x.cols(); ++j)
        for (int i = 0; i < H_x.rows(); ++i) H_xf(i, j) = H_x(i, j);
      for (int j = 0; j < H_f.cols(); ++j)
        for (int i = 0; i < H_x.rows(); ++i) H_xf(i, H_x.cols() + j) = H_f(i, j);
    }
    pre_rows[l] = H_xf.rows();
    pre_cols[l] = H_xf.cols();
    for (int j = 0; j < H_xf.cols(); ++j)
      for (int i = 0; i < H_xf.rows(); ++i) pre_H.push_back(H_xf(i, j));
    for (int i = 0; i < res.rows(); ++i) pre_H.push_back(res(i));
    for (const auto &v : Hx_order)
      for (int k = 0; k < v->size(); ++k) pre_ids.push_back(v->id() + k);
    for (int k = 0; k < landmark->size(); ++k) pre_ids.push_back(landmark->id() + k);
    any_pre = true;
  }
  ovp_slam_batch sb;
  memset(&sb, 0, sizeof(sb));
  sb.n_landmarks = L;
  sb.max_meas = M;
  sb.uv = uv.data();
  sb.clone_idx = cidx.data();
  sb.n_meas = nm.data();
  sb.p_FinG = pv.data();
  sb.p_FinG_fej = pf.data();
  sb.landmark_id = lmid.data();
  if (any_plane) {
    sb.plane_state_id = psid.data();
    sb.cp = cpv.data();
    sb.cp_fej = cpf.data();
  }
  if (any_pre) {
    sb.pre_rows = pre_rows.data();
    sb.pre_cols = pre_cols.data();
    sb.pre_H = pre_H.data();
    sb.pre_ids = pre_ids.data();
  }
  ovp_update_opts uo;
  memset(&uo, 0, sizeof(uo));
  uo.sigma_px = _options_slam.sigma_pix;
  uo.chi2_multiplier = _options_slam.chi2_multipler;
  uo.sigma_constraint = sigma_c;
  uo.do_fej = state->_options.do_fej ? 1 : 0;
  uo.do_calib_camera_pose = state->_options.do_calib_camera_pose ? 1 : 0;
  uo.do_calib_camera_intrinsics = state->_options.do_calib_camera_intrinsics ? 1 : 0;
  const int n = ovp_cov_size(state->_gpu);
  std::vector<double> dx(n, 0.0);
  std::vector<uint8_t> status(L, 0);
  ovp_update_info info;
  const int rc = ovp_slam_update(state->_gpu, &uo, &sb, dx.data(), status.data(), nullptr, &info);
  if (rc == OVP_E_NEGDIAG) {
    fprintf(stderr, "StateHelper::EKFUpdate() - negative covariance diagonal\n");
    std::exit(EXIT_FAILURE);
  }
  if (rc == OVP_E_CAPACITY) {  // the gate kernel's LDS bound: nothing was touched, the dense form takes the batch
    update_dense(state, feature_vec, feat2plane);
    return;
  }
  gpu_check2(rc, "ovp_slam_update");
  // :547-624 side effects of the gate, in the order of the vector
  size_t l = 0;
  auto it2 = feature_vec.begin();
  while (it2 != feature_vec.end()) {
    const uint8_t st = status[l];
    const size_t pid = planeid[l];
    ++l;
    if (st == 2 || (st == 0 && pid != 0)) state->_features_SLAM_to_PLANE[(*it2)->featid] = 0;  // :551 (set before the second test)
    if (st == 0) {  // :596-619
      state->_features_SLAM.at((*it2)->featid)->should_marg = true;
      (*it2)->to_delete = true;
      it2 = feature_vec.erase(it2);
      continue;
    }
    if (st == 1 && pid != 0) state->_features_SLAM_to_PLANE[(*it2)->featid] = pid;  // :623-624
    it2++;
  }
  for (size_t f = 0; f < feature_vec.size(); f++) feature_vec[f]->to_delete = true;  // :657-659
  if (info.n_accepted > 0) StateHelper::apply_correction(state, dx.data());  // :673 Type::update of every variable
}

bool UpdaterSLAM::_force_dense = false;

namespace {
// one landmark's linearised measurement: [H_x | H_landmark] over `order` (the landmark last), residual
struct DenseBlock {
  MatrixXd H;
  VectorXd r;
  std::vector<std::shared_ptr<Type>> order;
};

// r^T (H P H^T + I)^-1 r with P the marginal of the block's variables, by a Cholesky solve; +inf when S is not positive definite
double dense_block_chi2(std::shared_ptr<State> state, const DenseBlock &b) {
  const MatrixXd Pm = StateHelper::get_marginal_covariance(state, b.order);
  const int m = b.H.rows(), c = b.H.cols();
  std::vector<double> HP((size_t)m * c, 0.0), S((size_t)m * m, 0.0), y(m);
  for (int j = 0; j < c; ++j)
    for (int k = 0; k < c; ++k) {
      const double pkj = Pm(k, j);
      if (pkj == 0.0) continue;
      for (int i = 0; i < m; ++i) HP[(size_t)j * m + i] += b.H(i, k) * pkj;
    }
  for (int a = 0; a < m; ++a)
    for (int i = a; i < m; ++i) {
      double v = (i == a) ? 1.0 : 0.0;
      for (int j = 0; j < c; ++j) v += HP[(size_t)j * m + i] * b.H(a, j);
      S[(size_t)a * m + i] = v;  // lower triangle, column-major
    }
  double chi2 = 0.0;
  for (int j = 0; j < m; ++j) {  // left-looking factorization fused with the forward substitution of r
    double d = S[(size_t)j * m + j];
    for (int k = 0; k < j; ++k) d -= S[(size_t)k * m + j] * S[(size_t)k * m + j];
    if (!(d > 0.0)) return INFINITY;
    d = std::sqrt(d);
    S[(size_t)j * m + j] = d;
    double rj = b.r(j);
    for (int k = 0; k < j; ++k) rj -= S[(size_t)k * m + j] * y[k];
    y[j] = rj / d;
    chi2 += y[j] * y[j];
    for (int i = j + 1; i < m; ++i) {
      double v = S[(size_t)j * m + i];
      for (int k = 0; k < j; ++k) v -= S[(size_t)k * m + i] * S[(size_t)k * m + j];
      S[(size_t)j * m + i] = v / d;
    }
  }
  return chi2;
}
}  // namespace

void UpdaterSLAM::update_dense(std::shared_ptr<State> state, std::vector<std::shared_ptr<ov_core::Feature>> &feature_vec,
                               const std::map<size_t, size_t> &feat2plane) {
  typedef LandmarkRepresentation LR;
  const double sigma_c = state->_options.sigma_constraint, mult = _options_slam.chi2_multipler;
  // the landmark at its current estimate in its own representation, with or without the point-on-plane rows (update/UpdaterSLAM.cpp:478-522)
  auto linearize = [&](const ov_core::Feature &ft, const std::shared_ptr<Landmark> &lm, size_t planeid) {
    UpdaterHelper::UpdaterHelperFeature hf;
    hf.featid = ft.featid;
    hf.uvs = ft.uvs;
    hf.timestamps = ft.timestamps;
    hf.cam_ids = ft.cam_ids;
    const bool single = lm->_feat_representation == LR::ANCHORED_INVERSE_DEPTH_SINGLE;
    hf.feat_representation = single ? LR::ANCHORED_MSCKF_INVERSE_DEPTH : lm->_feat_representation;
    if (LR::is_relative_representation(hf.feat_representation)) {
      hf.anchor_cam_id = lm->_anchor_cam_id;
      hf.anchor_clone_timestamp = lm->_anchor_clone_timestamp;
      lm->get_xyz(false, hf.p_FinA);
      lm->get_xyz(true, hf.p_FinA_fej);
    } else {
      lm->get_xyz(false, hf.p_FinG);
      lm->get_xyz(true, hf.p_FinG_fej);
    }
    if (planeid != 0) {
      hf.planeid = planeid;
      const auto pl = state->_features_PLANE.at(planeid);
      for (int k = 0; k < 3; ++k) hf.cp_FinG[k] = pl->value()(k), hf.cp_FinG_fej[k] = pl->fej()(k);
    }
    MatrixXd H_f, H_x;
    DenseBlock b;
    UpdaterHelper::get_feature_jacobian_full(state, hf, _options_slam.sigma_pix, sigma_c, H_f, H_x, b.r, b.order);
    const int keep_f = single ? 1 : H_f.cols(), first_f = H_f.cols() - keep_f;  // single: only the depth column stays a variable
    MatrixXd H(H_x.rows(), H_x.cols() + keep_f);
    for (int i = 0; i < H_x.rows(); ++i) {
      for (int j = 0; j < H_x.cols(); ++j) H(i, j) = H_x(i, j);
      for (int j = 0; j < keep_f; ++j) H(i, H_x.cols() + j) = H_f(i, first_f + j);
    }
    if (single) {  // :499-515 the bearing columns are projected out
      MatrixXd H_b = H_f.block(0, 0, H_f.rows(), first_f);
      UpdaterHelper::nullspace_project_inplace(H_b, H, b.r);
    }
    b.H = H;
    b.order.push_back(lm);
    return b;
  };
  std::vector<DenseBlock> taken;
  std::vector<std::shared_ptr<ov_core::Feature>> survivors;
  for (auto &fp : feature_vec) {
    ov_core::Feature &ft = *fp;
    const auto lm = state->_features_SLAM.at(ft.featid);
    size_t pid = 0;
    {  // :465-475 a plane in the state that this landmark has not been taken off
      const auto it = feat2plane.find(ft.featid);
      const auto bad = state->_features_SLAM_to_PLANE.find(ft.featid);
      if (state->_options.use_plane_constraint && state->_options.use_plane_constraint_slamu && it != feat2plane.end() &&
          state->_features_PLANE.count(it->second) && !(bad != state->_features_SLAM_to_PLANE.end() && bad->second == 0))
        pid = it->second;
    }
    if (pid != 0 && lm->_feat_representation != LR::GLOBAL_3D) {
      fprintf(stderr, "UpdaterSLAM::update() - point-on-plane rows need a GLOBAL_3D landmark\n");
      std::exit(EXIT_FAILURE);
    }
    DenseBlock blk = linearize(ft, lm, pid);
    bool pass = dense_block_chi2(state, blk) <= mult * ovp_chi2_quantile_095(blk.r.rows());
    if (!pass && pid != 0) {  // :547-609 once more without the plane; the landmark is taken off it either way
      state->_features_SLAM_to_PLANE[ft.featid] = 0;
      pid = 0;
      blk = linearize(ft, lm, 0);
      pass = dense_block_chi2(state, blk) <= mult * ovp_chi2_quantile_095(blk.r.rows());
    }
    ft.to_delete = true;  // rejected: :612, kept: :657-659
    if (!pass) {
      lm->should_marg = true;
      continue;
    }
    if (pid != 0) state->_features_SLAM_to_PLANE[ft.featid] = pid;
    taken.push_back(blk);
    survivors.push_back(fp);
  }
  feature_vec = survivors;
  if (taken.empty()) return;
  // column layout of the stacked system: variables in order of first appearance (:634-646)
  std::vector<std::shared_ptr<Type>> big_order;
  std::map<const Type *, int> col_of;
  int n_cols = 0, n_rows = 0;
  for (const auto &b : taken) {
    n_rows += b.r.rows();
    for (const auto &v : b.order)
      if (!col_of.count(v.get())) {
        col_of[v.get()] = n_cols;
        big_order.push_back(v);
        n_cols += v->size();
      }
  }
  MatrixXd H_big = MatrixXd::Zero(n_rows, n_cols), R_big = MatrixXd::Identity(n_rows, n_rows);
  VectorXd r_big = VectorXd::Zero(n_rows, 1);
  int row0 = 0;
  for (const auto &b : taken) {
    int src = 0;
    for (const auto &v : b.order) {
      const int dst = col_of.at(v.get());
      for (int j = 0; j < v->size(); ++j)
        for (int i = 0; i < b.H.rows(); ++i) H_big(row0 + i, dst + j) = b.H(i, src + j);
      src += v->size();
    }
    for (int i = 0; i < b.r.rows(); ++i) r_big(row0 + i) = b.r(i);
    row0 += b.r.rows();
  }
  StateHelper::EKFUpdate(state, big_order, H_big, r_big, R_big);  // :673
}

// update/UpdaterSLAM.cpp:120-166 (the same block opens UpdaterMSCKF::update and UpdaterPlane::init_vio_plane): features that
// arrive with normalised measurements are triangulated (+ refined) on the device against the clone window; failures are flagged
// and leave the vector.  Features whose position was handed over (no uvs_norm) pass through.
void UpdaterSLAM::triangulate_on_device(std::shared_ptr<State> state, const ov_core::FeatureInitializerOptions &fio,
                                        std::vector<std::shared_ptr<ov_core::Feature>> &feature_vec) {
  bool any_norm = false;
  for (auto &f : feature_vec) any_norm = any_norm || (!f->uvs_norm.empty() && f->uvs_norm.size() == f->uvs.size());
  if (!any_norm) return;
  std::map<double, int> clone_slot;
  std::vector<std::shared_ptr<PoseJPL>> clones;
  UpdaterSLAM::upload_state_tables(state, clone_slot, clones);
  const int C = (int)clones.size(), F = (int)feature_vec.size();
  int M = 1;
  for (auto &f : feature_vec) M = std::max(M, (int)f->timestamps.size());
  std::vector<float> uv((size_t)F * M * 2, 0.f), uvn((size_t)F * M * 2, 0.f);
  std::vector<int> cidx((size_t)F * M, -1), nm(F);
  std::vector<double> pf((size_t)F * 3);
  for (int f = 0; f < F; ++f) {
    nm[f] = (int)feature_vec[f]->timestamps.size();
    for (int k = 0; k < nm[f]; ++k) {
      cidx[(size_t)f * M + k] = clone_slot.at(feature_vec[f]->timestamps[k]);
      uv[((size_t)f * M + k) * 2] = feature_vec[f]->uvs[2 * k];
      uv[((size_t)f * M + k) * 2 + 1] = feature_vec[f]->uvs[2 * k + 1];
    }
    for (size_t k = 0; k < feature_vec[f]->uvs_norm.size(); ++k) uvn[(size_t)f * M * 2 + k] = feature_vec[f]->uvs_norm[k];
    memcpy(&pf[3 * f], feature_vec[f]->p_FinG, 3 * sizeof(double));
  }
  ovp_feature_batch fb{F, M, uv.data(), cidx.data(), nm.data(), pf.data()};
  gpu_check2(ovp_batch_upload(state->_gpu, &fb), "ovp_batch_upload");
  ovp_triang_opts to;
  to.refine_features = fio.refine_features ? 1 : 0;
  to.triangulate_1d = fio.triangulate_1d ? 1 : 0;
  to.reserved = 0;
  to.max_runs = fio.max_runs;
  to.init_lamda = fio.init_lamda;
  to.max_lamda = fio.max_lamda;
  to.min_dx = fio.min_dx;
  to.min_dcost = fio.min_dcost;
  to.lam_mult = fio.lam_mult;
  to.min_dist = fio.min_dist;
  to.max_dist = fio.max_dist;
  to.max_baseline = fio.max_baseline;
  to.max_cond_number = fio.max_cond_number;
  std::vector<uint8_t> okv(F, 0);
  gpu_check2(ovp_triangulate(state->_gpu, &to, uvn.data(), pf.data(), okv.data()), "ovp_triangulate");
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
}

// OVP_HOST_PROFILE=1: wall-clock split of UpdaterSLAM::delayed_init, printed when the process exits
namespace {
struct DelayedInitProfile {
  double t_tri = 0, t_jac = 0, t_init = 0;
  long calls = 0, cands = 0, accepted = 0;
  bool on = getenv("OVP_HOST_PROFILE") != nullptr;
  ~DelayedInitProfile() {
    if (on && calls)
      fprintf(stderr, "[delayed_init] calls %ld candidates %ld accepted %ld | per call: triangulate %.1f us, jacobians %.1f us, initialize %.1f us (%.1f us per candidate)\n",
              calls, cands, accepted, 1e6 * t_tri / calls, 1e6 * t_jac / calls, 1e6 * t_init / calls, cands ? 1e6 * t_init / cands : 0.0);
  }
};
DelayedInitProfile g_diprof;
inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace

static bool fused_initialize_on() {  // OVP_HOST_INIT_SPLIT=1: the three separate device calls per candidate (A/B runs, tests)
  const char *e = getenv("OVP_HOST_INIT_SPLIT");
  return !(e && e[0] == '1');
}

void UpdaterSLAM::delayed_init(std::shared_ptr<State> state, std::vector<std::shared_ptr<ov_core::Feature>> &feature_vec,
                               const std::map<size_t, size_t> &feat2plane) {
  if (feature_vec.empty()) return;
  // :80-118 clean, need >= 2 measurements
  auto it0 = feature_vec.begin();
  while (it0 != feature_vec.end()) {
    clean_old_measurements(**it0, state->_clones_IMU);
    if ((*it0)->timestamps.size() < 2) {
      (*it0)->to_delete = true;
      it0 = feature_vec.erase(it0);
    } else {
      it0++;
    }
  }
  // :120-166 triangulation of the features that come with normalised measurements (the joint point / plane refinement
  // :168-202 is not repeated here: positions handed over, or triangulated, are used as they are)
  const double t_a = now_s();
  triangulate_on_device(state, _featinit, feature_vec);
  g_diprof.t_tri += now_s() - t_a;
  g_diprof.calls++;
  if (feature_vec.empty()) return;
  // a p_FinG that came without its anchor gets what ext single_triangulation would have left behind, with the poses as they
  // are now (the initialisations below move them)
  for (auto &fp : feature_vec) {
    ov_core::Feature &ft = *fp;
    if (ft.anchor_cam_id != -1) continue;
    ft.anchor_cam_id = 0;
    ft.anchor_clone_timestamp = ft.timestamps.back();
    auto an = state->_clones_IMU.at(ft.anchor_clone_timestamp);
    auto cal = state->_calib_IMUtoCAM.at(0);
    const double d[3] = {ft.p_FinG[0] - an->pos()[0], ft.p_FinG[1] - an->pos()[1], ft.p_FinG[2] - an->pos()[2]};
    double t[3];
    m3v(an->Rot(), d, t);
    m3v(cal->Rot(), t, ft.p_FinA);
    for (int k = 0; k < 3; ++k) ft.p_FinA[k] += cal->pos()[k];
  }
  // ---- the candidate loop on the device (ovp_slam_delayed_init, csrc/k_dinit.hip): GLOBAL_3D landmarks without plane rows -
  // every shipped configuration's feat_rep_slam - run as ONE enqueue: rows at the device tables, split, gate, augmentation,
  // update and the Type::update of the device tables per candidate without the host in between.  Candidates that carry plane
  // rows (use_plane_constraint_slamd with the plane in the state) and the other representations take the per-candidate path
  // below (dense Jacobians on the host); the order of the vector is kept (every initialisation moves the state of the next).
  auto wants_plane = [&](const ov_core::Feature &ft) {
    if (!(state->_options.use_plane_constraint && state->_options.use_plane_constraint_slamd)) return false;
    auto fp2 = feat2plane.find(ft.featid);
    if (fp2 == feat2plane.end() || state->_features_PLANE.find(fp2->second) == state->_features_PLANE.end()) return false;
    auto s2p = state->_features_SLAM_to_PLANE.find(ft.featid);
    return s2p == state->_features_SLAM_to_PLANE.end() || s2p->second != 0;
  };
  static const bool no_device_loop = getenv("OVP_HOST_DINIT_LOOP") != nullptr;  // A/B: the per-candidate path for everything
  const bool device_rep = state->_options.feat_rep_slam == LandmarkRepresentation::Representation::GLOBAL_3D && !no_device_loop &&
                          fused_initialize_on();
  // flushes the run [run_begin, it) of device candidates; returns the iterator behind the run (erased entries accounted for)
  auto flush_run = [&](size_t run_begin, size_t run_end) -> size_t {
    const int L = (int)(run_end - run_begin);
    if (L <= 0) return run_end;
    std::map<double, int> clone_slot;
    std::vector<std::shared_ptr<PoseJPL>> clones;
    upload_state_tables(state, clone_slot, clones);
    int M = 2;
    bool other_camera = false;
    for (int l = 0; l < L; ++l) {
      M = std::max(M, (int)feature_vec[run_begin + l]->timestamps.size());
      other_camera = other_camera || !feature_vec[run_begin + l]->only_camera0();
    }
    if (M > OVP_MAX_MEAS || other_camera) return (size_t)-1;  // (the per-candidate host loop has neither limit)
    std::vector<float> uv((size_t)L * M * 2, 0.f);
    std::vector<int> cidx((size_t)L * M, -1), nm(L);
    std::vector<double> pf((size_t)L * 3);
    for (int l = 0; l < L; ++l) {
      ov_core::Feature &ft = *feature_vec[run_begin + l];
      nm[l] = (int)ft.timestamps.size();
      for (int k = 0; k < nm[l]; ++k) {
        cidx[(size_t)l * M + k] = clone_slot.at(ft.timestamps[k]);
        uv[((size_t)l * M + k) * 2] = ft.uvs[2 * k];
        uv[((size_t)l * M + k) * 2 + 1] = ft.uvs[2 * k + 1];
      }
      memcpy(&pf[3 * l], ft.p_FinG, 3 * sizeof(double));
    }
    ovp_feature_batch fb{L, M, uv.data(), cidx.data(), nm.data(), pf.data()};
    ovp_update_opts uo;
    memset(&uo, 0, sizeof(uo));
    uo.sigma_px = _options_slam.sigma_pix;
    uo.chi2_multiplier = _options_slam.chi2_multipler;
    uo.sigma_constraint = state->_options.sigma_constraint;
    uo.do_fej = state->_options.do_fej ? 1 : 0;
    uo.do_calib_camera_pose = state->_options.do_calib_camera_pose ? 1 : 0;
    uo.do_calib_camera_intrinsics = state->_options.do_calib_camera_intrinsics ? 1 : 0;
    const int stride = ovp_cov_size(state->_gpu) + 3 * L;
    std::vector<uint8_t> okv(L, 0);
    std::vector<int> nid(L, -1);
    std::vector<double> dl((size_t)3 * L, 0.0), dxs((size_t)L * stride, 0.0);
    const double t_c = now_s();
    const int rc = ovp_slam_delayed_init(state->_gpu, &uo, &fb, okv.data(), nullptr, nid.data(), dl.data(), dxs.data(), stride);
    if (rc == OVP_E_CAPACITY) return (size_t)-1;  // nothing was touched: the caller takes these candidates one by one
    if (rc == OVP_E_NEGDIAG) {
      fprintf(stderr, "StateHelper::EKFUpdate() - negative covariance diagonal\n");
      std::exit(EXIT_FAILURE);
    }
    gpu_check2(rc, "ovp_slam_delayed_init");
    g_diprof.t_init += now_s() - t_c;
    g_diprof.cands += L;
    size_t pos = run_begin;
    for (int l = 0; l < L; ++l) {
      ov_core::Feature &ft = *feature_vec[pos];
      ft.to_delete = true;
      if (!okv[l]) {  // :360-363
        feature_vec.erase(feature_vec.begin() + (long)pos);
        continue;
      }
      auto landmark = std::make_shared<Landmark>(3);  // :285-296
      landmark->_featid = ft.featid;
      landmark->_feat_representation = LandmarkRepresentation::Representation::GLOBAL_3D;
      landmark->_unique_camera_id = ft.anchor_cam_id;
      landmark->set_from_xyz(ft.p_FinG, false);
      landmark->set_from_xyz(ft.p_FinG, true);
      VectorXd d(3, 1);
      for (int k = 0; k < 3; ++k) d(k) = dl[3 * l + k];
      landmark->update(d);  // state/StateHelper.cpp:577
      landmark->set_local_id(nid[l]);
      state->_variables.push_back(landmark);
      StateHelper::apply_correction(state, &dxs[(size_t)l * stride]);  // :483-485 Type::update of every variable
      state->_features_SLAM.insert({ft.featid, landmark});
      g_diprof.accepted++;
      ++pos;
    }
    return pos;
  };
  if (device_rep) {
    size_t i = 0;
    bool all_done = true;
    while (i < feature_vec.size()) {
      size_t j = i;
      while (j < feature_vec.size() && !wants_plane(*feature_vec[j])) ++j;
      if (j > i) {
        const size_t behind = flush_run(i, j);
        if (behind == (size_t)-1) {
          all_done = false;
          break;
        }
        i = behind;
      }
      if (i < feature_vec.size() && wants_plane(*feature_vec[i])) {
        all_done = false;  // a candidate with plane rows: the per-candidate path takes over from here
        break;
      }
    }
    if (all_done) return;
    // (what is left - from position i on - goes through the loop below; the accepted ones in front of it are done and flagged)
    std::vector<std::shared_ptr<ov_core::Feature>> rest(feature_vec.begin() + (long)i, feature_vec.end());
    std::vector<std::shared_ptr<ov_core::Feature>> head(feature_vec.begin(), feature_vec.begin() + (long)i);
    delayed_init_host_loop(state, rest, feat2plane);
    feature_vec = head;
    feature_vec.insert(feature_vec.end(), rest.begin(), rest.end());
    return;
  }
  delayed_init_host_loop(state, feature_vec, feat2plane);
}

// update/UpdaterSLAM.cpp:204-364 candidate by candidate: dense Jacobians on the host (every representation, plane rows with their
// fallback), StateHelper::initialize on the device per candidate
void UpdaterSLAM::delayed_init_host_loop(std::shared_ptr<State> state, std::vector<std::shared_ptr<ov_core::Feature>> &feature_vec,
                                         const std::map<size_t, size_t> &feat2plane) {
  typedef LandmarkRepresentation LR;
  const auto rep_state = state->_options.feat_rep_slam;
  // :230-246 the single inverse depth is linearised as the MSCKF inverse depth, its depth column moved to the state side
  const bool single = rep_state == LR::ANCHORED_INVERSE_DEPTH_SINGLE;
  const auto rep_lin = single ? LR::ANCHORED_MSCKF_INVERSE_DEPTH : rep_state;
  const bool anchored = LR::is_relative_representation(rep_lin);
  const double sigma_c = state->_options.sigma_constraint, mult = _options_slam.chi2_multipler;
  // the plane a candidate may be initialised on: in the state, and not one this feature was taken off before (:210-228)
  auto plane_of = [&](size_t featid) -> size_t {
    if (!(state->_options.use_plane_constraint && state->_options.use_plane_constraint_slamd)) return 0;
    const auto it = feat2plane.find(featid);
    if (it == feat2plane.end() || !state->_features_PLANE.count(it->second)) return 0;
    const auto off = state->_features_SLAM_to_PLANE.find(featid);
    return (off != state->_features_SLAM_to_PLANE.end() && off->second == 0) ? 0 : it->second;
  };
  std::vector<std::shared_ptr<ov_core::Feature>> survivors;
  for (auto &fp : feature_vec) {
    ov_core::Feature &ft = *fp;
    UpdaterHelper::UpdaterHelperFeature feat;
    feat.featid = ft.featid;
    feat.uvs = ft.uvs;
    feat.timestamps = ft.timestamps;
    feat.cam_ids = ft.cam_ids;
    feat.feat_representation = rep_lin;
    double *value = anchored ? feat.p_FinA : feat.p_FinG, *first = anchored ? feat.p_FinA_fej : feat.p_FinG_fej;
    memcpy(value, anchored ? ft.p_FinA : ft.p_FinG, 3 * sizeof(double));
    memcpy(first, value, 3 * sizeof(double));  // a new landmark's first estimate is its value
    if (anchored) {
      feat.anchor_cam_id = ft.anchor_cam_id;
      feat.anchor_clone_timestamp = ft.anchor_clone_timestamp;
    }
    auto landmark = std::make_shared<Landmark>(single ? 1 : 3);  // :285-296
    landmark->_featid = ft.featid;
    landmark->_feat_representation = rep_state;
    landmark->_unique_camera_id = ft.anchor_cam_id;
    if (anchored) {
      landmark->_anchor_cam_id = ft.anchor_cam_id;
      landmark->_anchor_clone_timestamp = ft.anchor_clone_timestamp;
    }
    landmark->set_from_xyz(value, false);
    landmark->set_from_xyz(first, true);
    // with the plane's rows first, then - when that fails the gate - once more without them (:304-359)
    size_t pid = plane_of(ft.featid);
    bool in_state = false;
    for (int attempt = 0; attempt < 2 && !in_state; ++attempt) {
      feat.planeid = pid;
      if (pid != 0) {
        const auto pl = state->_features_PLANE.at(pid);
        for (int k = 0; k < 3; ++k) feat.cp_FinG[k] = pl->value()(k), feat.cp_FinG_fej[k] = pl->fej()(k);
      }
      MatrixXd H_f, H_x;
      VectorXd res;
      std::vector<std::shared_ptr<Type>> order;
      const double t_b = now_s();
      UpdaterHelper::get_feature_jacobian_full(state, feat, _options_slam.sigma_pix, sigma_c, H_f, H_x, res, order);
      if (single) {  // :262-283 the depth column joins the state side, the bearing columns are projected out
        MatrixXd H_xd(H_x.rows(), H_x.cols() + 1);
        for (int i = 0; i < H_x.rows(); ++i) {
          for (int j = 0; j < H_x.cols(); ++j) H_xd(i, j) = H_x(i, j);
          H_xd(i, H_x.cols()) = H_f(i, H_f.cols() - 1);
        }
        MatrixXd H_bearing = H_f.block(0, 0, H_f.rows(), H_f.cols() - 1);
        UpdaterHelper::nullspace_project_inplace(H_bearing, H_xd, res);
        H_x = H_xd.block(0, 0, H_xd.rows(), H_xd.cols() - 1);
        H_f = H_xd.block(0, H_xd.cols() - 1, H_xd.rows(), 1);
      }
      g_diprof.t_jac += now_s() - t_b;
      if (attempt == 0) g_diprof.cands++;
      MatrixXd R = MatrixXd::Identity(res.rows(), res.rows());
      const double t_c = now_s();
      in_state = StateHelper::initialize(state, landmark, order, H_x, H_f, R, res, mult);
      g_diprof.t_init += now_s() - t_c;
      if (in_state || pid == 0) break;
      state->_features_SLAM_to_PLANE[ft.featid] = 0;  // the plane is dropped for this feature, whatever the second attempt says
      pid = 0;
    }
    g_diprof.accepted += in_state;
    ft.to_delete = true;
    if (!in_state) continue;
    state->_features_SLAM.insert({ft.featid, landmark});
    if (pid != 0) state->_features_SLAM_to_PLANE[ft.featid] = pid;
    survivors.push_back(fp);
  }
  feature_vec = survivors;
}

// ---- update/UpdaterPlane.cpp ---------------------------------------------------------------------
// ---- update/UpdaterSLAM.cpp:684-850 ---------------------------------------------------------------
void UpdaterSLAM::change_anchors(std::shared_ptr<State> state) {
  if ((int)state->_clones_IMU.size() <= state->_options.max_clone_size) return;  // :687-689
  const double marg_timestep = state->margtimestep();
  for (auto &f : state->_features_SLAM) {
    if (!LandmarkRepresentation::is_relative_representation(f.second->_feat_representation)) continue;  // :697-699
    assert(marg_timestep <= f.second->_anchor_clone_timestamp);
    if (f.second->_anchor_clone_timestamp == marg_timestep)
      perform_anchor_change(state, f.second, state->_timestamp, (size_t)f.second->_anchor_cam_id);
  }
}

void UpdaterSLAM::perform_anchor_change(std::shared_ptr<State> state, std::shared_ptr<Landmark> landmark, double new_anchor_timestamp,
                                        size_t new_cam_id) {
  assert(LandmarkRepresentation::is_relative_representation(landmark->_feat_representation));
  assert(landmark->_anchor_cam_id != -1);
  // :716-727 Jacobians of p_FinG w.r.t. the old representation
  UpdaterHelper::UpdaterHelperFeature old_feat;
  old_feat.featid = landmark->_featid;
  old_feat.feat_representation = landmark->_feat_representation;
  old_feat.anchor_cam_id = landmark->_anchor_cam_id;
  old_feat.anchor_clone_timestamp = landmark->_anchor_clone_timestamp;
  landmark->get_xyz(false, old_feat.p_FinA);
  landmark->get_xyz(true, old_feat.p_FinA_fej);
  MatrixXd H_f_old;
  std::vector<MatrixXd> H_x_old;
  std::vector<std::shared_ptr<Type>> x_order_old;
  UpdaterHelper::get_feature_jacobian_representation(state, old_feat, H_f_old, H_x_old, x_order_old);
  // :730-734
  UpdaterHelper::UpdaterHelperFeature new_feat;
  new_feat.featid = landmark->_featid;
  new_feat.feat_representation = landmark->_feat_representation;
  new_feat.anchor_cam_id = (int)new_cam_id;
  new_feat.anchor_clone_timestamp = new_anchor_timestamp;
  // :739-775 the landmark in the new anchor camera frame, at the current and at the first estimates
  auto transfer = [&](bool fej, const double p_old[3], double p_new[3]) {
    auto co = state->_clones_IMU.at(old_feat.anchor_clone_timestamp), cn = state->_clones_IMU.at(new_feat.anchor_clone_timestamp);
    auto ko = state->_calib_IMUtoCAM.at(old_feat.anchor_cam_id), kn = state->_calib_IMUtoCAM.at(new_feat.anchor_cam_id);
    double R_GtoOLD[9], R_GtoNEW[9], p_OLDinG[3], p_NEWinG[3];
    m3m(ko->Rot(), fej ? co->Rot_fej() : co->Rot(), R_GtoOLD);
    m3m(kn->Rot(), fej ? cn->Rot_fej() : cn->Rot(), R_GtoNEW);
    const double *po = fej ? co->pos_fej() : co->pos(), *pn = fej ? cn->pos_fej() : cn->pos();
    for (int i = 0; i < 3; ++i) {
      p_OLDinG[i] = po[i] - (R_GtoOLD[i] * ko->pos()[0] + R_GtoOLD[3 + i] * ko->pos()[1] + R_GtoOLD[6 + i] * ko->pos()[2]);
      p_NEWinG[i] = pn[i] - (R_GtoNEW[i] * kn->pos()[0] + R_GtoNEW[3 + i] * kn->pos()[1] + R_GtoNEW[6 + i] * kn->pos()[2]);
    }
    // p_new = R_GtoNEW (R_GtoOLD^T p_old + p_OLDinG - p_NEWinG)
    double g[3];
    for (int i = 0; i < 3; ++i)
      g[i] = R_GtoOLD[i] * p_old[0] + R_GtoOLD[3 + i] * p_old[1] + R_GtoOLD[6 + i] * p_old[2] + p_OLDinG[i] - p_NEWinG[i];
    m3v(R_GtoNEW, g, p_new);
  };
  transfer(false, old_feat.p_FinA, new_feat.p_FinA);
  transfer(true, old_feat.p_FinA_fej, new_feat.p_FinA_fej);
  // :778-781
  MatrixXd H_f_new;
  std::vector<MatrixXd> H_x_new;
  std::vector<std::shared_ptr<Type>> x_order_new;
  UpdaterHelper::get_feature_jacobian_representation(state, new_feat, H_f_new, H_x_new, x_order_new);
  // :787-808 order of the old states the new landmark error depends on
  std::vector<std::shared_ptr<Type>> phi_order_NEW{landmark}, phi_order_OLD;
  std::vector<std::pair<std::shared_ptr<Type>, int>> Phi_id_map;
  auto find_phi = [&](const std::shared_ptr<Type> &v) {
    for (auto &p : Phi_id_map)
      if (p.first == v) return p.second;
    return -1;
  };
  int current_it = 0;
  for (const auto &var : x_order_old)
    if (find_phi(var) < 0) {
      Phi_id_map.push_back({var, current_it});
      phi_order_OLD.push_back(var);
      current_it += var->size();
    }
  for (const auto &var : x_order_new)
    if (find_phi(var) < 0) {
      Phi_id_map.push_back({var, current_it});
      phi_order_OLD.push_back(var);
      current_it += var->size();
    }
  Phi_id_map.push_back({landmark, current_it});
  phi_order_OLD.push_back(landmark);
  current_it += landmark->size();
  // :811-836  pf_new_error = Hfnew^-1 (Hfold pf_olderror + Hxold x_olderror - Hxnew x_newerror)
  const int phisize = (new_feat.feat_representation != LandmarkRepresentation::ANCHORED_INVERSE_DEPTH_SINGLE) ? 3 : 1;
  MatrixXd Phi = MatrixXd::Zero(phisize, current_it), Q = MatrixXd::Zero(phisize, phisize);
  MatrixXd H_f_new_inv(phisize, 3);
  if (phisize == 1) {
    double nn = 0;
    for (int k = 0; k < 3; ++k) nn += H_f_new(k, 0) * H_f_new(k, 0);
    for (int k = 0; k < 3; ++k) H_f_new_inv(0, k) = H_f_new(k, 0) / nn;
  } else {
    // 3x3 inverse (the reference solves with a column-pivoted QR)
    const MatrixXd &A = H_f_new;
    const double c00 = A(1, 1) * A(2, 2) - A(1, 2) * A(2, 1), c01 = A(0, 2) * A(2, 1) - A(0, 1) * A(2, 2), c02 = A(0, 1) * A(1, 2) - A(0, 2) * A(1, 1);
    const double det = A(0, 0) * c00 + A(1, 0) * c01 + A(2, 0) * c02;
    H_f_new_inv(0, 0) = c00 / det;
    H_f_new_inv(0, 1) = c01 / det;
    H_f_new_inv(0, 2) = c02 / det;
    H_f_new_inv(1, 0) = (A(1, 2) * A(2, 0) - A(1, 0) * A(2, 2)) / det;
    H_f_new_inv(1, 1) = (A(0, 0) * A(2, 2) - A(0, 2) * A(2, 0)) / det;
    H_f_new_inv(1, 2) = (A(0, 2) * A(1, 0) - A(0, 0) * A(1, 2)) / det;
    H_f_new_inv(2, 0) = (A(1, 0) * A(2, 1) - A(1, 1) * A(2, 0)) / det;
    H_f_new_inv(2, 1) = (A(0, 1) * A(2, 0) - A(0, 0) * A(2, 1)) / det;
    H_f_new_inv(2, 2) = (A(0, 0) * A(1, 1) - A(0, 1) * A(1, 0)) / det;
  }
  auto add_block = [&](int col0, const MatrixXd &B, double sign) {
    for (int i = 0; i < phisize; ++i)
      for (int j = 0; j < B.cols(); ++j) {
        double a = 0;
        for (int k = 0; k < 3; ++k) a += H_f_new_inv(i, k) * B(k, j);
        Phi(i, col0 + j) += sign * a;
      }
  };
  for (size_t i = 0; i < H_x_old.size(); i++) add_block(find_phi(x_order_old[i]), H_x_old[i], 1.0);
  add_block(find_phi(landmark), H_f_old, 1.0);
  for (size_t i = 0; i < H_x_new.size(); i++) add_block(find_phi(x_order_new[i]), H_x_new[i], -1.0);
  StateHelper::EKFPropagation(state, phi_order_NEW, phi_order_OLD, Phi, Q);  // :839
  // :842-848
  landmark->_anchor_cam_id = new_feat.anchor_cam_id;
  landmark->_anchor_clone_timestamp = new_feat.anchor_clone_timestamp;
  landmark->set_from_xyz(new_feat.p_FinA, false);
  landmark->set_from_xyz(new_feat.p_FinA_fej, true);
  landmark->has_had_anchor_change = true;
}

UpdaterPlane::UpdaterPlane(UpdaterOptions &options, ov_core::FeatureInitializerOptions &feat_init_options)
    : _options(options), _featinit(feat_init_options) {
  _options.sigma_pix_sq = std::pow(_options.sigma_pix, 2);
}

void UpdaterPlane::nullspace_project_inplace(MatrixXd &H_f, MatrixXd &H_x, MatrixXd &H_cp, VectorXd &res) {
  assert(H_f.rows() >= H_f.cols());
  for (int n = 0; n < H_f.cols(); ++n)
    for (int m = H_f.rows() - 1; m > n; m--) {
      double c, s;
      givens(H_f(m - 1, n), H_f(m, n), c, s);
      rot2(H_f, m - 1, n, c, s);
      rot2(H_x, m - 1, 0, c, s);
      rot2(H_cp, m - 1, 0, c, s);
      rot2(res, m - 1, 0, c, s);
    }
  const int k = H_f.cols();
  H_x = H_x.block(k, 0, H_x.rows() - k, H_x.cols());
  H_cp = H_cp.block(k, 0, H_cp.rows() - k, H_cp.cols());
  res = res.block(k, 0, res.rows() - k, res.cols());
}

void UpdaterPlane::measurement_compress_inplace(MatrixXd &H_x, MatrixXd &H_cp, VectorXd &res) {
  if (H_x.rows() <= H_x.cols()) return;
  for (int n = 0; n < H_x.cols(); n++)
    for (int m = H_x.rows() - 1; m > n; m--) {
      double c, s;
      givens(H_x(m - 1, n), H_x(m, n), c, s);
      rot2(H_x, m - 1, n, c, s);
      rot2(H_cp, m - 1, 0, c, s);
      rot2(res, m - 1, 0, c, s);
    }
  const int r = std::min(H_x.rows(), H_x.cols());
  H_x = H_x.block(0, 0, r, H_x.cols());
  H_cp = H_cp.block(0, 0, r, H_cp.cols());
  res = res.block(0, 0, r, 1);
}

void UpdaterPlane::init_vio_plane(std::shared_ptr<State> state, std::vector<std::shared_ptr<ov_core::Feature>> &feature_vec,
                                  std::vector<std::shared_ptr<ov_core::Feature>> &feature_vec_used,
                                  const std::map<size_t, size_t> &feat2plane) {
  if (feature_vec.empty() || feat2plane.empty()) return;
  std::map<double, int> clone_slot;
  std::vector<std::shared_ptr<PoseJPL>> clones;
  UpdaterSLAM::upload_state_tables(state, clone_slot, clones);
  auto calib = state->_calib_IMUtoCAM.at(0);
  auto intr = state->_cam_intrinsics.at(0);
  const int C = (int)clones.size();
  auto upload = [&](const std::vector<std::shared_ptr<ov_core::Feature>> &fv, int &M_out) {
    const int Fn = (int)fv.size();
    int Mx = 1;
    for (auto &f : fv) Mx = std::max(Mx, (int)f->timestamps.size());
    std::vector<float> uvb((size_t)Fn * Mx * 2, 0.f);
    std::vector<int> cidxb((size_t)Fn * Mx, -1), nmb(Fn);
    std::vector<double> pfb((size_t)Fn * 3);
    for (int f = 0; f < Fn; ++f) {
      nmb[f] = (int)fv[f]->timestamps.size();
      for (int k = 0; k < nmb[f]; ++k) {
        cidxb[(size_t)f * Mx + k] = clone_slot.at(fv[f]->timestamps[k]);
        uvb[((size_t)f * Mx + k) * 2] = fv[f]->uvs[2 * k];
        uvb[((size_t)f * Mx + k) * 2 + 1] = fv[f]->uvs[2 * k + 1];
      }
      memcpy(&pfb[3 * f], fv[f]->p_FinG, 3 * sizeof(double));
    }
    ovp_feature_batch fbb{Fn, Mx, uvb.data(), cidxb.data(), nmb.data(), pfb.data()};
    gpu_check2(ovp_batch_upload(state->_gpu, &fbb), "ovp_batch_upload");
    M_out = Mx;
  };

  // planes to initialise, ascending id (:297 iterates a std::map), and the features each of them may use
  std::vector<size_t> todo;
  std::set<size_t> fit_kept;  // features that survived plane_fitting / optimize_plane (fit path only)
  bool fitted = false;
  if (!state->_plane_estimates_cp_inG.empty()) {
    // pre-fitted entry point: the caller ran the fit (or owns better estimates)
    for (const auto &pe : state->_plane_estimates_cp_inG)
      if (state->_features_PLANE.find(pe.first) == state->_features_PLANE.end()) todo.push_back(pe.first);
    for (auto &f : feature_vec) clean_old_measurements(*f, state->_clones_IMU);
  } else {
    fitted = true;
    // :76-107 on-plane features of planes that are not in the state, with at least two measurements in the window
    std::vector<std::shared_ptr<ov_core::Feature>> valid;
    auto it0 = feature_vec.begin();
    while (it0 != feature_vec.end()) {
      auto fp = feat2plane.find((*it0)->featid);
      if (fp == feat2plane.end() || state->_features_PLANE.count(fp->second)) {
        clean_old_measurements(**it0, state->_clones_IMU);  // (the batch below needs window measurements only)
        it0++;
        continue;
      }
      clean_old_measurements(**it0, state->_clones_IMU);
      if ((*it0)->timestamps.size() < 2) {
        it0 = feature_vec.erase(it0);
      } else {
        valid.push_back(*it0);
        it0++;
      }
    }
    // :126-149 triangulate + refine; failures only leave the candidate set
    bool any_norm = false;
    for (auto &f : valid) any_norm = any_norm || (!f->uvs_norm.empty() && f->uvs_norm.size() == f->uvs.size());
    if (any_norm && !valid.empty()) {
      int Mv = 1;
      upload(valid, Mv);
      const int Fv = (int)valid.size();
      std::vector<float> uvn((size_t)Fv * Mv * 2, 0.f);
      for (int f = 0; f < Fv; ++f)
        for (size_t k = 0; k < valid[f]->uvs_norm.size(); ++k) uvn[(size_t)f * Mv * 2 + k] = valid[f]->uvs_norm[k];
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
      std::vector<double> pfv((size_t)Fv * 3);
      std::vector<uint8_t> okv(Fv, 0);
      gpu_check2(ovp_triangulate(state->_gpu, &to, uvn.data(), pfv.data(), okv.data()), "ovp_triangulate");
      std::vector<std::shared_ptr<ov_core::Feature>> good;
      for (int f = 0; f < Fv; ++f) {
        const bool had_norm = !valid[f]->uvs_norm.empty();
        if (had_norm && !okv[f]) continue;
        if (had_norm) memcpy(valid[f]->p_FinG, &pfv[3 * f], 3 * sizeof(double));
        good.push_back(valid[f]);
      }
      valid = good;
    }
    // :167-176 shortest tracks first (std::sort like the reference: the order of equal tracks is libstdc++'s)
    std::sort(valid.begin(), valid.end(), [](const std::shared_ptr<ov_core::Feature> &a, const std::shared_ptr<ov_core::Feature> &b) {
      return a->timestamps.size() < b->timestamps.size();
    });
    // :179-196
    std::map<size_t, size_t> plane_feat_count;
    std::map<size_t, std::vector<std::shared_ptr<ov_core::Feature>>> plane_feats;
    for (auto &feat : valid) {
      const size_t planeid = feat2plane.at(feat->featid);
      if ((int)plane_feat_count[planeid] > state->_options.max_msckf_plane) continue;
      plane_feat_count[planeid]++;
      plane_feats[planeid].push_back(feat);
    }
    // :221-290 initial guess of every plane: RANSAC fit, then joint refinement with its features
    PlaneFitting::ClonesCam clones_cam;
    for (const auto &cl : state->_clones_IMU) {
      PlaneFitting::ClonePose cpose;
      const double *Ri = cl.second->Rot(), *Rc = calib->Rot(), *pi = cl.second->pos(), *pc = calib->pos();
      for (int i = 0; i < 3; ++i)
        for (int k = 0; k < 3; ++k) cpose.R[3 * i + k] = Rc[3 * i] * Ri[k] + Rc[3 * i + 1] * Ri[3 + k] + Rc[3 * i + 2] * Ri[6 + k];
      for (int i = 0; i < 3; ++i) cpose.p[i] = pi[i] - (cpose.R[i] * pc[0] + cpose.R[3 + i] * pc[1] + cpose.R[6 + i] * pc[2]);
      clones_cam[0][cl.first] = cpose;
    }
    const double sigma_px_norm = _options.sigma_pix / intr->value()(0);
    double stateI[7], calib0[7];
    memcpy(stateI, state->_imu->quat(), 4 * sizeof(double));
    memcpy(stateI + 4, state->_imu->pos(), 3 * sizeof(double));
    memcpy(calib0, calib->quat(), 4 * sizeof(double));
    memcpy(calib0 + 4, calib->pos(), 3 * sizeof(double));
    for (auto &fp : plane_feats) {
      bool all_norm = true;
      for (auto &ft : fp.second) all_norm = all_norm && (ft->uvs_norm.size() == 2 * ft->timestamps.size());
      if (!all_norm) continue;  // optimize_plane needs the normalised measurements
      double abcd[4];
      if (!PlaneFitting::plane_fitting(fp.second, abcd, state->_options.plane_init_min_feat, state->_options.plane_init_max_cond)) continue;
      double cp0[3] = {-abcd[0] * abcd[3], -abcd[1] * abcd[3], -abcd[2] * abcd[3]};
      if (!PlaneFitting::optimize_plane(fp.second, cp0, clones_cam, sigma_px_norm, state->_options.sigma_constraint, false, stateI, calib0))
        continue;
      state->_plane_estimates_cp_inG[fp.first] = {cp0[0], cp0[1], cp0[2]};
      todo.push_back(fp.first);
      for (auto &ft : fp.second) fit_kept.insert(ft->featid);
    }
  }
  if (todo.empty()) {
    if (fitted) state->_plane_estimates_cp_inG.clear();
    return;
  }
  const int F = (int)feature_vec.size();
  int M = 1;
  upload(feature_vec, M);
  std::vector<int> pof(F, 0);
  for (int f = 0; f < F; ++f) {
    auto it = feat2plane.find(feature_vec[f]->featid);
    if (it != feat2plane.end()) {
      auto pos = std::find(todo.begin(), todo.end(), it->second);
      if (pos != todo.end() && (!fitted || fit_kept.count(feature_vec[f]->featid))) pof[f] = 1 + (int)(pos - todo.begin());
    }
  }
  const int NP = (int)todo.size();
  std::vector<double> cpv(3 * NP), cpn(3 * NP);
  std::vector<int> sid(NP, -1), nid(NP, -1);
  for (int k = 0; k < NP; ++k)
    for (int a = 0; a < 3; ++a) cpv[3 * k + a] = state->_plane_estimates_cp_inG.at(todo[k])[a];
  ovp_update_opts o{_options.sigma_pix,
                    _options.chi2_multipler,
                    state->_options.sigma_constraint,
                    state->_options.do_fej ? 1 : 0,
                    state->_options.do_calib_camera_pose ? 1 : 0,
                    state->_options.do_calib_camera_intrinsics ? 1 : 0,
                    0};
  const int stride = state->_options.max_state_size;
  std::vector<double> dxp((size_t)NP * stride, 0.0);
  std::vector<uint8_t> pok(NP, 0), fused(F, 0);
  ovp_plane_batch pb{NP, pof.data(), cpv.data(), cpv.data(), sid.data()};
  gpu_check2(ovp_plane_init(state->_gpu, &o, &pb, state->_options.const_init_multi, state->_options.const_init_chi2, dxp.data(), stride,
                            pok.data(), nullptr, nullptr, nid.data(), cpn.data(), fused.data()),
             "ovp_plane_init");
  for (int k = 0; k < NP; ++k) {
    if (!pok[k]) continue;
    // Type::update of the variables that existed before this plane, then the plane itself joins the state (:441-449)
    for (auto &var : state->_variables) {
      VectorXd d(var->size(), 1);
      for (int a = 0; a < var->size(); ++a) d(a) = dxp[(size_t)k * stride + var->id() + a];
      var->update(d);
    }
    auto plane = std::make_shared<Vec>(3);
    VectorXd v(3, 1), vf(3, 1);
    for (int a = 0; a < 3; ++a) {
      v(a) = cpn[3 * k + a];
      vf(a) = cpv[3 * k + a];
    }
    plane->set_value(v);
    plane->set_fej(vf);
    plane->set_local_id(nid[k]);
    state->_variables.push_back(plane);
    state->_features_PLANE.insert({todo[k], plane});
  }
  if (fitted) state->_plane_estimates_cp_inG.clear();  // the fit path's estimates are local to this call (:220)
  // :459-475 features consumed by an initialised plane leave the MSCKF vector
  auto it = feature_vec.begin();
  size_t f = 0;
  while (it != feature_vec.end()) {
    if (fused[f]) {
      (*it)->to_delete = true;
      feature_vec_used.push_back(*it);
      it = feature_vec.erase(it);
    } else {
      it++;
    }
    ++f;
  }
}

}  // namespace ov_plane
