// Host mirror of track_plane/PlaneFitting.{h,cpp} over the C-ABI (ovp_plane_fitting / ovp_plane_optimize): same static
// signatures, same side effects on the feature vectors (feats becomes the inlier set, feat->p_FinG is overwritten for the
// kept features, cp_inG only on success).
#include "ov_plane_host.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define PRINT_ERROR(...) fprintf(stderr, __VA_ARGS__)

namespace ov_plane {

ovp_ctx *PlaneFitting::_gpu = nullptr;
int PlaneFitting::_variant = 0;

static void pf_check(int rc, const char *what) {
  if (rc != 0) {
    PRINT_ERROR("PlaneFitting: %s failed (%d)\n", what, rc);
    std::exit(EXIT_FAILURE);
  }
}

// PlaneFitting.cpp:42-82.  The five-point / inlier-set solves of plane_fitting run on the device; this standalone form is the
// same least-squares problem on the host for callers that fit a handful of points (track_plane/TrackPlane.cpp, out of scope).
bool PlaneFitting::fit_plane(const std::vector<std::shared_ptr<ov_core::Feature>> &feats, double abcd[4], double cond_thresh,
                             bool cond_check) {
  if (feats.size() < 3) return false;
  double M[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, sv[3] = {0, 0, 0};
  for (auto &f : feats)
    for (int a = 0; a < 3; ++a) {
      sv[a] += f->p_FinG[a];
      for (int b = 0; b < 3; ++b) M[3 * a + b] += f->p_FinG[a] * f->p_FinG[b];
    }
  if (cond_check) {  // cond(A) = sqrt(lambda_max / lambda_min) of A^T A (cyclic Jacobi)
    double a[9];
    memcpy(a, M, sizeof(a));
    for (int sweep = 0; sweep < 30; ++sweep) {
      if (a[1] * a[1] + a[2] * a[2] + a[5] * a[5] < 1e-300) break;
      for (int p = 0; p < 2; ++p)
        for (int q = p + 1; q < 3; ++q) {
          const double apq = a[3 * p + q];
          if (apq == 0.0) continue;
          const double theta = (a[3 * q + q] - a[3 * p + p]) / (2.0 * apq);
          const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
          const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
          for (int k = 0; k < 3; ++k) {
            const double akp = a[3 * k + p], akq = a[3 * k + q];
            a[3 * k + p] = c * akp - s * akq;
            a[3 * k + q] = s * akp + c * akq;
          }
          for (int k = 0; k < 3; ++k) {
            const double apk = a[3 * p + k], aqk = a[3 * q + k];
            a[3 * p + k] = c * apk - s * aqk;
            a[3 * q + k] = s * apk + c * aqk;
          }
        }
    }
    const double lo = std::min(a[0], std::min(a[4], a[8])), hi = std::max(a[0], std::max(a[4], a[8]));
    if (std::sqrt(std::max(hi, 0.0)) / std::sqrt(std::max(lo, 0.0)) > cond_thresh) return false;
  }
  // Cholesky of the normal equations
  const double l00 = std::sqrt(M[0]), l10 = M[3] / l00, l20 = M[6] / l00;
  const double l11 = std::sqrt(M[4] - l10 * l10), l21 = (M[7] - l20 * l10) / l11;
  const double l22 = std::sqrt(M[8] - l20 * l20 - l21 * l21);
  const double y0 = -sv[0] / l00, y1 = (-sv[1] - l10 * y0) / l11, y2 = (-sv[2] - l20 * y0 - l21 * y1) / l22;
  double x[3];
  x[2] = y2 / l22;
  x[1] = (y1 - l21 * x[2]) / l11;
  x[0] = (y0 - l10 * x[1] - l20 * x[2]) / l00;
  const double nn = std::sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
  if (!(nn > 0.0) || !std::isfinite(nn)) return false;
  for (int k = 0; k < 3; ++k) abcd[k] = x[k] / nn;
  abcd[3] = 1.0 / nn;
  return std::fabs(abcd[3]) > 0.02;  // |cp| = |d| for a unit normal (:77-80)
}

// PlaneFitting.cpp:84-199
bool PlaneFitting::plane_fitting(std::vector<std::shared_ptr<ov_core::Feature>> &feats, double plane_abcd[4], int min_inlier_num,
                                 double max_plane_solver_condition_number) {
  if (!_gpu) {
    PRINT_ERROR("PlaneFitting::plane_fitting() - no device context (construct a State first)\n");
    std::exit(EXIT_FAILURE);
  }
  const int n = (int)feats.size();
  if (n == 0) return false;
  std::vector<double> pts(3 * (size_t)n);
  for (int i = 0; i < n; ++i) memcpy(&pts[3 * i], feats[i]->p_FinG, 3 * sizeof(double));
  const int fs[2] = {0, n};
  ovp_planefit_batch b{1, fs, pts.data(), min_inlier_num, max_plane_solver_condition_number, _variant};
  std::vector<uint8_t> inl(n, 0);
  uint8_t ok = 0;
  double abcd[4];
  pf_check(ovp_plane_fitting(_gpu, &b, abcd, inl.data(), &ok), "ovp_plane_fitting");
  if (!ok) return false;
  memcpy(plane_abcd, abcd, sizeof(abcd));
  std::vector<std::shared_ptr<ov_core::Feature>> best;
  for (int i = 0; i < n; ++i)
    if (inl[i]) best.push_back(feats[i]);
  feats = best;  // :190
  return true;
}

// PlaneFitting.cpp:201-514
bool PlaneFitting::optimize_plane(std::vector<std::shared_ptr<ov_core::Feature>> &feats, double cp_inG[3], ClonesCam &clonesCAM,
                                  double sigma_px_norm, double sigma_c, bool fix_plane, const double stateI[7], const double calib0[7]) {
  if (!_gpu) {
    PRINT_ERROR("PlaneFitting::optimize_plane() - no device context (construct a State first)\n");
    std::exit(EXIT_FAILURE);
  }
  const int nf = (int)feats.size();
  if ((!fix_plane && nf < 4) || (fix_plane && nf == 0)) return false;  // :214-217
  std::vector<double> p0(3 * (size_t)nf), uv, Rc, pc;
  std::vector<int> obs_start(nf), n_obs(nf);
  for (int f = 0; f < nf; ++f) {
    memcpy(&p0[3 * f], feats[f]->p_FinG, 3 * sizeof(double));
    obs_start[f] = (int)(uv.size() / 2);
    n_obs[f] = (int)feats[f]->timestamps.size();
    if (feats[f]->uvs_norm.size() != 2 * feats[f]->timestamps.size()) {
      PRINT_ERROR("PlaneFitting::optimize_plane() - feature %zu has no normalised measurements\n", feats[f]->featid);
      std::exit(EXIT_FAILURE);
    }
    for (int k = 0; k < n_obs[f]; ++k) {
      const ClonePose &cl = clonesCAM.at(0).at(feats[f]->timestamps[k]);
      uv.push_back((double)feats[f]->uvs_norm[2 * k]);
      uv.push_back((double)feats[f]->uvs_norm[2 * k + 1]);
      Rc.insert(Rc.end(), cl.R, cl.R + 9);
      pc.insert(pc.end(), cl.p, cl.p + 3);
    }
  }
  const int fs[2] = {0, nf};
  const uint8_t fix = fix_plane ? 1 : 0;
  ovp_planeopt_batch b;
  b.n_planes = 1;
  b.feat_start = fs;
  b.p_FinG = p0.data();
  b.obs_start = obs_start.data();
  b.n_obs = n_obs.data();
  b.n_obs_total = (int)(uv.size() / 2);
  b.uv_norm = uv.data();
  b.R_GtoC = Rc.data();
  b.p_CinG = pc.data();
  b.cp = cp_inG;
  b.fix_plane = &fix;
  b.sigma_px_norm = sigma_px_norm;
  b.sigma_c = sigma_c;
  ov_type::quat_2_Rot(stateI, b.R_GtoI);
  memcpy(b.p_IinG, stateI + 4, 3 * sizeof(double));
  ov_type::quat_2_Rot(calib0, b.R_ItoC);
  memcpy(b.p_IinC, calib0 + 4, 3 * sizeof(double));
  double cp_out[3];
  std::vector<double> p_out(3 * (size_t)nf);
  std::vector<uint8_t> kept(nf, 0);
  uint8_t ok = 0;
  int its = 0;
  pf_check(ovp_plane_optimize(_gpu, &b, cp_out, p_out.data(), kept.data(), &ok, &its), "ovp_plane_optimize");
  if (!ok) return false;
  memcpy(cp_inG, cp_out, sizeof(cp_out));  // :432-436
  std::vector<std::shared_ptr<ov_core::Feature>> inliers;
  for (int f = 0; f < nf; ++f)
    if (kept[f]) {
      memcpy(feats[f]->p_FinG, &p_out[3 * f], 3 * sizeof(double));  // :481
      inliers.push_back(feats[f]);
    }
  feats = inliers;  // :511
  return true;
}

}  // namespace ov_plane
