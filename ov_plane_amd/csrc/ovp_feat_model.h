// Measurement model of one bearing row (shared by the point and the plane feature kernels).
// update/UpdaterHelper.cpp:345-444 (GLOBAL_3D, radtan or equidistant lens, mono): residual, H_f row, clone block, calibration block.
#pragma once
#include "ovp_dev.h"
#include "ovp_kernels.h"

namespace ovp {

// lane <-> (observation a, row r); ci = clone slot of the observation.  Invalid lanes return zeros.
// LM_FEJ: the feature is a landmark of the state with a first estimate of its own (pf_fej, update/UpdaterHelper.cpp:300-303, :381);
// MSCKF features linearise at their value (fej == value, update/UpdaterMSCKF.cpp:499-500) and take the default instantiation.
template <bool LM_FEJ = false>
__device__ __forceinline__ void build_bearing_row(const FeatParams& p, int f, int a, int r, bool valid, int ci,
                                                  double (&jrow)[6], double (&crow)[14], double (&hf)[3], double& res,
                                                  const double* pf_fej = nullptr) {
  const double* __restrict__ cal = p.cal;  // [0..8] R_ItoC, [9..11] p_IinC, [12..19] intrinsics (device memory)
  const double* Rc = cal;
  const double pIC0 = cal[9], pIC1 = cal[10], pIC2 = cal[11];
    const double pf0 = p.p_FinG[3 * f], pf1 = p.p_FinG[3 * f + 1], pf2 = p.p_FinG[3 * f + 2];
    const double* R = p.clone_R + 9 * ci;
    const double* pp = p.clone_p + 3 * ci;
    double d0 = pf0 - pp[0], d1 = pf1 - pp[1], d2 = pf2 - pp[2];
    double pI0 = R[0] * d0 + R[1] * d1 + R[2] * d2;
    double pI1 = R[3] * d0 + R[4] * d1 + R[5] * d2;
    double pI2 = R[6] * d0 + R[7] * d1 + R[8] * d2;
    double pC0 = Rc[0] * pI0 + Rc[1] * pI1 + Rc[2] * pI2 + pIC0;
    double pC1 = Rc[3] * pI0 + Rc[4] * pI1 + Rc[5] * pI2 + pIC1;
    double pC2 = Rc[6] * pI0 + Rc[7] * pI1 + Rc[8] * pI2 + pIC2;
    const double x = pC0 / pC2, y = pC1 / pC2;
    // ext CamRadtan::distort_d / CamEqui::distort_d (call site UpdaterHelper.cpp:365); the model is wave-uniform
    const double fx = cal[12 + 0], fy = cal[12 + 1], cx = cal[12 + 2], cy = cal[12 + 3];
    const double k1 = cal[12 + 4], k2 = cal[12 + 5], p1 = cal[12 + 6], p2 = cal[12 + 7];
    const double r2 = x * x + y * y, r4 = r2 * r2;
    double g = 0.0, x1, y1;
    // equidistant model: theta powers, 1/r and the 2x2 d(xy1)/d(xy) (ext CamEqui::compute_distort_jacobian)
    double th3 = 0.0, th5 = 0.0, th7 = 0.0, th9 = 0.0, inv_r = 1.0, e00 = 0.0, e01 = 0.0, e11 = 0.0;
    if (p.fisheye) {
      const double rr = sqrt(r2);
      const double th = atan(rr), t2 = th * th;
      th3 = th * t2;
      th5 = th3 * t2;
      th7 = th5 * t2;
      th9 = th7 * t2;
      const double th_d = th + k1 * th3 + k2 * th5 + p1 * th7 + p2 * th9;
      inv_r = (rr > 1e-8) ? 1.0 / rr : 1.0;
      const double cdist = (rr > 1e-8) ? th_d * inv_r : 1.0;
      x1 = x * cdist;
      y1 = y * cdist;
      const double dthd_dth = 1.0 + 3.0 * k1 * t2 + 5.0 * k2 * t2 * t2 + 7.0 * p1 * t2 * t2 * t2 + 9.0 * p2 * t2 * t2 * t2 * t2;
      const double dth_dr = 1.0 / (r2 + 1.0);
      const double a0 = -x * th_d * inv_r * inv_r + x * inv_r * dthd_dth * dth_dr;
      const double a1 = -y * th_d * inv_r * inv_r + y * inv_r * dthd_dth * dth_dr;
      e00 = th_d * inv_r + a0 * (x * inv_r);
      e01 = a0 * (y * inv_r);  // = a1 * (x * inv_r)
      e11 = th_d * inv_r + a1 * (y * inv_r);
    } else {
      g = 1.0 + k1 * r2 + k2 * r4;
      x1 = x * g + 2.0 * p1 * x * y + p2 * (r2 + 2.0 * x * x);
      y1 = y * g + p1 * (r2 + 2.0 * y * y) + 2.0 * p2 * x * y;
    }
    const double ud = fx * x1 + cx, vd = fy * y1 + cy;
    const float* uvp = p.uv + ((size_t)f * p.max_meas + (valid ? a : 0)) * 2;
    const double um = (double)uvp[0], vm = (double)uvp[1];
    res = p.white_px * (r ? (vm - vd) : (um - ud));
    // FEJ re-evaluation :376-385
    if (p.do_fej) {
      R = p.clone_R_fej + 9 * ci;
      pp = p.clone_p_fej + 3 * ci;
      if constexpr (LM_FEJ) {
        d0 = pf_fej[0] - pp[0];
        d1 = pf_fej[1] - pp[1];
        d2 = pf_fej[2] - pp[2];
      } else {
        d0 = pf0 - pp[0];
        d1 = pf1 - pp[1];
        d2 = pf2 - pp[2];
      }
      pI0 = R[0] * d0 + R[1] * d1 + R[2] * d2;
      pI1 = R[3] * d0 + R[4] * d1 + R[5] * d2;
      pI2 = R[6] * d0 + R[7] * d1 + R[8] * d2;
      pC0 = Rc[0] * pI0 + Rc[1] * pI1 + Rc[2] * pI2 + pIC0;
      pC1 = Rc[3] * pI0 + Rc[4] * pI1 + Rc[5] * pI2 + pIC1;
      pC2 = Rc[6] * pI0 + Rc[7] * pI1 + Rc[8] * pI2 + pIC2;
    }
    // ext Cam*::compute_distort_jacobian at the non-FEJ uv_norm (:383,389), row r only
    double dzn0, dzn1;  // dz_dzn[r][0..1]
    if (p.fisheye) {
      if (r == 0) {
        dzn0 = fx * e00;
        dzn1 = fx * e01;
        crow[6] = x1;
        crow[7] = 0.0;
        crow[8] = 1.0;
        crow[9] = 0.0;
        const double q = fx * x * inv_r;
        crow[10] = q * th3;
        crow[11] = q * th5;
        crow[12] = q * th7;
        crow[13] = q * th9;
      } else {
        dzn0 = fy * e01;
        dzn1 = fy * e11;
        crow[6] = 0.0;
        crow[7] = y1;
        crow[8] = 0.0;
        crow[9] = 1.0;
        const double q = fy * y * inv_r;
        crow[10] = q * th3;
        crow[11] = q * th5;
        crow[12] = q * th7;
        crow[13] = q * th9;
      }
    } else if (r == 0) {
      dzn0 = fx * (g + 2.0 * k1 * x * x + 4.0 * k2 * x * x * r2 + 2.0 * p1 * y + 6.0 * p2 * x);
      dzn1 = fx * (2.0 * k1 * x * y + 4.0 * k2 * x * y * r2 + 2.0 * p1 * x + 2.0 * p2 * y);
      crow[6] = x1;
      crow[7] = 0.0;
      crow[8] = 1.0;
      crow[9] = 0.0;
      crow[10] = fx * x * r2;
      crow[11] = fx * x * r4;
      crow[12] = 2.0 * fx * x * y;
      crow[13] = fx * (r2 + 2.0 * x * x);
    } else {
      dzn0 = fy * (2.0 * k1 * x * y + 4.0 * k2 * x * y * r2 + 2.0 * p1 * x + 2.0 * p2 * y);
      dzn1 = fy * (g + 2.0 * k1 * y * y + 4.0 * k2 * y * y * r2 + 6.0 * p1 * y + 2.0 * p2 * x);
      crow[6] = 0.0;
      crow[7] = y1;
      crow[8] = 0.0;
      crow[9] = 1.0;
      crow[10] = fy * y * r2;
      crow[11] = fy * y * r4;
      crow[12] = fy * (r2 + 2.0 * y * y);
      crow[13] = 2.0 * fy * x * y;
    }
    // dzn_dpfc (:392-393) folded with dz_dzn (:407): dz_dpfc row r
    const double iz = 1.0 / pC2;
    const double z0 = dzn0 * iz, z1 = dzn1 * iz, z2 = -(dzn0 * pC0 + dzn1 * pC1) * iz * iz;
    const double w = p.white_px;
    // dpfc_dpfg = R_ItoC R_GtoIi (:396);  H_f row (:411)
    // v = (dz_dpfc row) * R_ItoC   (1x3)
    const double v0 = z0 * Rc[0] + z1 * Rc[3] + z2 * Rc[6];
    const double v1 = z0 * Rc[1] + z1 * Rc[4] + z2 * Rc[7];
    const double v2 = z0 * Rc[2] + z1 * Rc[5] + z2 * Rc[8];
    hf[0] = w * (v0 * R[0] + v1 * R[3] + v2 * R[6]);
    hf[1] = w * (v0 * R[1] + v1 * R[4] + v2 * R[7]);
    hf[2] = w * (v0 * R[2] + v1 * R[5] + v2 * R[8]);
    // clone block (:399-401,414): [ dz_dpfc R_ItoC skew(p_FinIi) , -dz_dpfg ]
    jrow[0] = w * (v1 * pI2 - v2 * pI1);
    jrow[1] = w * (v2 * pI0 - v0 * pI2);
    jrow[2] = w * (v0 * pI1 - v1 * pI0);
    jrow[3] = -hf[0];
    jrow[4] = -hf[1];
    jrow[5] = -hf[2];
    // extrinsics block (:426-435): [ dz_dpfc skew(p_FinCi - p_IinC) , dz_dpfc ]
    const double q0 = pC0 - pIC0, q1 = pC1 - pIC1, q2 = pC2 - pIC2;
    crow[0] = w * (z1 * q2 - z2 * q1);
    crow[1] = w * (z2 * q0 - z0 * q2);
    crow[2] = w * (z0 * q1 - z1 * q0);
    crow[3] = w * z0;
    crow[4] = w * z1;
    crow[5] = w * z2;
#pragma unroll
    for (int k = 6; k < 14; ++k) crow[k] *= w;  // intrinsics block (:438-440)
#pragma unroll
    for (int k = 0; k < 14; ++k)
      if (!((p.calmask >> k) & 1)) crow[k] = 0.0;
    if (!valid) {
      res = 0.0;
#pragma unroll
      for (int k = 0; k < 6; ++k) jrow[k] = 0.0;
#pragma unroll
      for (int k = 0; k < 14; ++k) crow[k] = 0.0;
      hf[0] = hf[1] = hf[2] = 0.0;
    }
  }

}  // namespace ovp
