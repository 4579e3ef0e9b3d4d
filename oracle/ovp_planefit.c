/*
 * TEST INFRASTRUCTURE ONLY  -  CPU restatement of rpng/ov_plane's plane fitting:
 *   PlaneFitting::fit_plane        track_plane/PlaneFitting.cpp:42-82
 *   PlaneFitting::plane_fitting    track_plane/PlaneFitting.cpp:84-199   (RANSAC, std::mt19937(8888) + std::shuffle)
 *   PlaneFitting::optimize_plane   track_plane/PlaneFitting.cpp:201-514  (Ceres DENSE_SCHUR + DOGLEG, CauchyLoss(1), <= 12 iterations)
 * with the factors  Factor_PointOnPlane (ceres/Factor_PointOnPlane.cpp:41-71) and ov_init::Factor_ImageReprojCalib.
 *
 * PARITY UNPINNED, and more so than the rest of the oracle: three libraries that are not in /root/reference decide the bits.
 *   - libstdc++'s std::shuffle / uniform_int_distribution: restated here in BOTH forms that exist - the rejection
 *     ("downscaling") form of GCC <= 10, which is what the reference's documented platforms (Ubuntu 18.04 / 20.04) link, and
 *     the Lemire form GCC >= 11 uses for 32-bit engines.  The form matching the local compiler is checked against the real
 *     std::shuffle in tests/test_planefit_cpu.py; std::mt19937 is checked against its 10000th value (4123659995).
 *   - Eigen: colPivHouseholderQr().solve / JacobiSVD singular values are restated as a column-pivoted Householder least
 *     squares and as the square roots of the eigenvalues of A^T A.
 *   - Ceres (1.13 / 1.14 on those platforms): the trust-region loop, the traditional dogleg strategy, Jacobi scaling, the
 *     robust-loss corrector and the termination tests are restated from the published algorithm (trust_region_minimizer.cc,
 *     dogleg_strategy.cc, corrector.cc) with Ceres' default options; ov_init::Factor_ImageReprojCalib (open_vins, absent)
 *     is the pinhole reprojection in normalised coordinates with identity extrinsics / intrinsics as optimize_plane sets
 *     them up (PlaneFitting.cpp:300-325).  Converged optima are solver independent; the accept / reject decision of a run
 *     that needs close to 12 iterations is not.
 * Only tests/ may load this.
 */
#include "ovp_planefit.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------------------------ */
/* std::mt19937 (ISO C++ [rand.eng.mers]): w=32 n=624 m=397 r=31 a=0x9908b0df u=11 s=7 b=0x9d2c5680 t=15 c=0xefc60000 l=18 */
/* ------------------------------------------------------------------------------------------------------------------ */
void ovo_mt_seed(ovo_mt19937 *g, uint32_t seed) {
  g->mt[0] = seed;
  for (int i = 1; i < 624; ++i) g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
  g->idx = 624;
}

uint32_t ovo_mt_next(ovo_mt19937 *g) {
  if (g->idx >= 624) {
    for (int i = 0; i < 624; ++i) {
      const uint32_t y = (g->mt[i] & 0x80000000u) | (g->mt[(i + 1) % 624] & 0x7fffffffu);
      g->mt[i] = g->mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    g->idx = 0;
  }
  uint32_t y = g->mt[g->idx++];
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}

/* std::uniform_int_distribution<unsigned long>{0, urange}(g) for a 32-bit engine (bits/uniform_int_dist.h).
 * variant 0: GCC <= 10 - scaling = 0xFFFFFFFF / (urange + 1), reject values >= (urange + 1) * scaling, divide.
 * variant 1: GCC >= 11 - Lemire's nearly divisionless method on 64-bit products (_S_nd). */
uint64_t ovo_uniform_int(ovo_mt19937 *g, uint64_t urange, int variant) {
  const uint64_t urngrange = 0xFFFFFFFFull;
  if (urange >= urngrange) return ovo_mt_next(g); /* not reached on this path (ranges are feature counts) */
  const uint64_t uerange = urange + 1;
  if (variant == 0) {
    const uint64_t scaling = urngrange / uerange;
    const uint64_t past = uerange * scaling;
    uint64_t ret;
    do ret = ovo_mt_next(g);
    while (ret >= past);
    return ret / scaling;
  }
  const uint32_t range = (uint32_t)uerange;
  uint64_t product = (uint64_t)ovo_mt_next(g) * (uint64_t)range;
  uint32_t low = (uint32_t)product;
  if (low < range) {
    const uint32_t threshold = (uint32_t)(-range) % range;
    while (low < threshold) {
      product = (uint64_t)ovo_mt_next(g) * (uint64_t)range;
      low = (uint32_t)product;
    }
  }
  return product >> 32;
}

/* std::shuffle (bits/stl_algo.h, GCC >= 7): two swap positions from one draw while range^2 fits the engine's range. */
void ovo_shuffle(int *v, int n, ovo_mt19937 *g, int variant) {
  if (n <= 0) return;
  const uint64_t urngrange = 0xFFFFFFFFull;
  const uint64_t urange = (uint64_t)n;
#define OVO_SWAP(a, b) \
  do {                 \
    int t_ = v[a];     \
    v[a] = v[b];       \
    v[b] = t_;         \
  } while (0)
  if (urngrange / urange >= urange) {
    int i = 1;
    if ((urange % 2) == 0) {
      const int pos = (int)ovo_uniform_int(g, 1, variant);
      OVO_SWAP(i, pos);
      ++i;
    }
    while (i != n) {
      const uint64_t swap_range = (uint64_t)i + 1;
      const uint64_t b1 = swap_range + 1;
      const uint64_t x = ovo_uniform_int(g, swap_range * b1 - 1, variant);
      const int p0 = (int)(x / b1), p1 = (int)(x % b1);
      OVO_SWAP(i, p0);
      ++i;
      OVO_SWAP(i, p1);
      ++i;
    }
    return;
  }
  for (int i = 1; i < n; ++i) {
    const int pos = (int)ovo_uniform_int(g, (uint64_t)i, variant);
    OVO_SWAP(i, pos);
  }
#undef OVO_SWAP
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* small dense helpers                                                                                                  */
/* ------------------------------------------------------------------------------------------------------------------ */
static double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static double norm3(const double *a) { return sqrt(dot3(a, a)); }

/* eigenvalues of a symmetric 3x3 (cyclic Jacobi); w ascending */
static void sym3_eigvals(const double S[9], double w[3]) {
  double a[9];
  memcpy(a, S, sizeof(a));
  for (int sweep = 0; sweep < 30; ++sweep) {
    const double off = a[1] * a[1] + a[2] * a[2] + a[5] * a[5];
    /* converged: the off-diagonal part is below the rounding of the diagonal (cyclic Jacobi: four or five sweeps; the eigenvalues
     * only feed the condition-number test of fit_plane) */
    if (off <= 1e-34 * (a[0] * a[0] + a[4] * a[4] + a[8] * a[8])) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        const double apq = a[3 * p + q];
        if (apq == 0.0) continue;
        const double theta = (a[3 * q + q] - a[3 * p + p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; ++k) { /* columns */
          const double akp = a[3 * k + p], akq = a[3 * k + q];
          a[3 * k + p] = c * akp - s * akq;
          a[3 * k + q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; ++k) { /* rows */
          const double apk = a[3 * p + k], aqk = a[3 * q + k];
          a[3 * p + k] = c * apk - s * aqk;
          a[3 * q + k] = s * apk + c * aqk;
        }
      }
  }
  w[0] = a[0];
  w[1] = a[4];
  w[2] = a[8];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2 - i; ++j)
      if (w[j] > w[j + 1]) {
        const double t = w[j];
        w[j] = w[j + 1];
        w[j + 1] = t;
      }
}

/* x = argmin |A x - b|, A n x 3 row-major, Householder QR with column pivoting (Eigen::ColPivHouseholderQR::solve) */
static void lsq3_colpiv(const double *A, const double *b, int n, double x[3]) {
  double *M = (double *)malloc(sizeof(double) * (size_t)n * 4);
  for (int i = 0; i < n; ++i) {
    M[4 * i + 0] = A[3 * i + 0];
    M[4 * i + 1] = A[3 * i + 1];
    M[4 * i + 2] = A[3 * i + 2];
    M[4 * i + 3] = b[i];
  }
  int perm[3] = {0, 1, 2};
  int rank = 0;
  double maxdiag = 0.0;
  for (int k = 0; k < 3 && k < n; ++k) {
    /* pivot: remaining column of largest norm */
    int best = k;
    double bestn = -1.0;
    for (int j = k; j < 3; ++j) {
      double s = 0.0;
      for (int i = k; i < n; ++i) s += M[4 * i + j] * M[4 * i + j];
      if (s > bestn) {
        bestn = s;
        best = j;
      }
    }
    if (best != k) {
      for (int i = 0; i < n; ++i) {
        const double t = M[4 * i + k];
        M[4 * i + k] = M[4 * i + best];
        M[4 * i + best] = t;
      }
      const int t = perm[k];
      perm[k] = perm[best];
      perm[best] = t;
    }
    /* Householder vector for column k, rows k.. */
    double sigma = 0.0;
    for (int i = k + 1; i < n; ++i) sigma += M[4 * i + k] * M[4 * i + k];
    const double c0 = M[4 * k + k];
    double beta, tau;
    if (sigma == 0.0) {
      beta = c0;
      tau = 0.0;
    } else {
      beta = sqrt(c0 * c0 + sigma);
      if (c0 >= 0.0) beta = -beta;
      for (int i = k + 1; i < n; ++i) M[4 * i + k] /= (c0 - beta);
      tau = (beta - c0) / beta;
    }
    /* apply H = I - tau v v^T (v_k = 1) to the remaining columns and the right-hand side */
    for (int j = k + 1; j < 4; ++j) {
      double s = M[4 * k + j];
      for (int i = k + 1; i < n; ++i) s += M[4 * i + k] * M[4 * i + j];
      s *= tau;
      M[4 * k + j] -= s;
      for (int i = k + 1; i < n; ++i) M[4 * i + j] -= s * M[4 * i + k];
    }
    M[4 * k + k] = beta;
    if (fabs(beta) > maxdiag) maxdiag = fabs(beta);
    ++rank;
  }
  /* rank threshold as Eigen: |R_kk| > eps * size * max|R_kk| */
  const double thr = 2.220446049250313e-16 * (double)(n < 3 ? n : 3) * maxdiag;
  int r = 0;
  for (int k = 0; k < rank; ++k)
    if (fabs(M[4 * k + k]) > thr) ++r;
  double y[3] = {0, 0, 0};
  for (int k = r - 1; k >= 0; --k) {
    double s = M[4 * k + 3];
    for (int j = k + 1; j < r; ++j) s -= M[4 * k + j] * y[j];
    y[k] = s / M[4 * k + k];
  }
  x[0] = x[1] = x[2] = 0.0;
  for (int k = 0; k < r; ++k) x[perm[k]] = y[k];
  free(M);
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* PlaneFitting::fit_plane                                                     track_plane/PlaneFitting.cpp:42-82      */
/* ------------------------------------------------------------------------------------------------------------------ */
int ovo_fit_plane(const double *pts, int n, double cond_thresh, int cond_check, double abcd[4]) {
  if (n < 3) return 0; /* :46-49 */
  if (cond_check) {    /* :59-66  JacobiSVD: sigma_i = sqrt(eig_i(A^T A)) */
    double S[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, w[3];
    for (int i = 0; i < n; ++i)
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) S[3 * a + b] += pts[3 * i + a] * pts[3 * i + b];
    sym3_eigvals(S, w);
    const double smax = sqrt(w[2] > 0 ? w[2] : 0.0), smin = sqrt(w[0] > 0 ? w[0] : 0.0);
    const double cond = smax / smin;
    if (cond > cond_thresh) return 0;
  }
  double *b = (double *)malloc(sizeof(double) * (size_t)n);
  for (int i = 0; i < n; ++i) b[i] = -1.0; /* :53 */
  double x[3];
  lsq3_colpiv(pts, b, n, x); /* :70 */
  free(b);
  const double nn = norm3(x);
  abcd[0] = x[0] / nn; /* :72-74 */
  abcd[1] = x[1] / nn;
  abcd[2] = x[2] / nn;
  abcd[3] = 1.0 / nn;
  const double cp[3] = {-abcd[0] * abcd[3], -abcd[1] * abcd[3], -abcd[2] * abcd[3]}; /* :77-80 */
  return norm3(cp) > 0.02;
}

static double pt_plane_dist(const double *p, const double *abcd) { return dot3(p, abcd) + abcd[3]; } /* PlaneFitting.h:73-75 */

/* ------------------------------------------------------------------------------------------------------------------ */
/* PlaneFitting::plane_fitting                                                 track_plane/PlaneFitting.cpp:84-199     */
/* ------------------------------------------------------------------------------------------------------------------ */
int ovo_plane_fitting(const double *pts, int n, int min_inlier_num, double max_cond, int variant, double abcd[4],
                      unsigned char *inlier, int *n_inliers) {
  const int ransac_n = 5, max_iter = 200; /* :88-93 */
  const double min_inlier_ratio = 0.80, max_err = 0.05, min_dist = 0.05;
  const int ratio_n = (int)((double)n * min_inlier_ratio);
  const size_t min_on_plane = (size_t)(min_inlier_num > ratio_n ? min_inlier_num : ratio_n);
  ovo_mt19937 g;
  ovo_mt_seed(&g, 8888u);
  for (int i = 0; i < n; ++i) inlier[i] = 0;
  *n_inliers = 0;
  if (n < min_inlier_num) return 0; /* :97-100 */

  int *order = (int *)malloc(sizeof(int) * (size_t)n);
  unsigned char *cur = (unsigned char *)malloc((size_t)n), *best = (unsigned char *)calloc((size_t)n, 1);
  double *sub = (double *)malloc(sizeof(double) * 3 * (size_t)(n > ransac_n ? n : ransac_n));
  double best_error = -1.0;
  size_t best_count = 0;
  int ok = 0;
  for (int it = 0; it < max_iter; ++it) {
    for (int i = 0; i < n; ++i) order[i] = i; /* :108 copy of feats */
    ovo_shuffle(order, n, &g, variant);       /* :110 */
    int set[5], ns = 0;                       /* :113-135 */
    for (int k = 0; k < n && ns < ransac_n; ++k) {
      const double *p = pts + 3 * order[k];
      int good = 1;
      for (int s = 0; s < ns; ++s) {
        const double *q = pts + 3 * set[s];
        const double d[3] = {q[0] - p[0], q[1] - p[1], q[2] - p[2]};
        if (norm3(d) < min_dist) {
          good = 0;
          break;
        }
      }
      if (ns == 0 || good) set[ns++] = order[k];
    }
    if (ns != ransac_n) goto done; /* :138-141: failure of the whole call */
    for (int s = 0; s < ransac_n; ++s) memcpy(sub + 3 * s, pts + 3 * set[s], sizeof(double) * 3);
    double cand[4];
    if (ovo_fit_plane(sub, ransac_n, max_cond, 1, cand)) { /* :144 */
      double avg = 0.0;
      size_t cnt = 0;
      for (int i = 0; i < n; ++i) { /* :147-155 */
        const double e = pt_plane_dist(pts + 3 * i, cand);
        cur[i] = fabs(e) < max_err;
        if (cur[i]) {
          ++cnt;
          avg += fabs(e);
        }
      }
      avg /= (double)cnt;
      const int valid = (cnt > min_on_plane && avg < max_err); /* :159-166 */
      const int better = (best_count < cnt) || (best_count == cnt && avg < best_error);
      if (valid && better) {
        memcpy(best, cur, (size_t)n);
        best_count = cnt;
        best_error = avg;
      }
    }
  }
  if (best_count > 0) { /* :171-192 */
    int m = 0;
    for (int i = 0; i < n; ++i)
      if (best[i]) memcpy(sub + 3 * (m++), pts + 3 * i, sizeof(double) * 3);
    if (ovo_fit_plane(sub, m, max_cond, 0, abcd)) {
      memcpy(inlier, best, (size_t)n);
      *n_inliers = m;
      ok = 1;
    }
  }
done:
  free(order);
  free(cur);
  free(best);
  free(sub);
  return ok;
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* PlaneFitting::optimize_plane                                                track_plane/PlaneFitting.cpp:201-514    */
/* ------------------------------------------------------------------------------------------------------------------ */
/* parameters: x = [p_f (3 per free feature) ... , cp (3, unless fixed)].  Residual blocks in the order of :261-371.    */
typedef struct {
  const ovo_planeopt_problem *pr;
  int nfree;       /* free features */
  int *slot;       /* feature -> parameter slot (or -1 when constant) */
  int cp_off;      /* offset of cp in x, -1 when fixed */
  int npar;
  int nres_rows;   /* residual rows that depend on a free parameter */
} popt_t;

/* One residual block evaluated at (p, cp): rows (1 or 2), whitened residual r, Jacobians Jp (rows x 3), Jc (rows x 3) */
static void reproj_block(const double *R, const double *pc, const double *uv, double sigma, const double *p, double r[2],
                         double Jp[6]) {
  /* ov_init::Factor_ImageReprojCalib with q_ItoC = identity, p_IinC = 0, intrinsics (1,1,0,0 | 0,0,0,0), radtan:
   *   p_FinCi = R_GtoCi (p_FinG - p_CiinG), uv = p_FinCi.xy / p_FinCi.z, r = (uv - uv_meas) / sigma_px_norm */
  const double d[3] = {p[0] - pc[0], p[1] - pc[1], p[2] - pc[2]};
  const double x = R[0] * d[0] + R[1] * d[1] + R[2] * d[2];
  const double y = R[3] * d[0] + R[4] * d[1] + R[5] * d[2];
  const double z = R[6] * d[0] + R[7] * d[1] + R[8] * d[2];
  const double w = 1.0 / sigma;
  r[0] = w * (x / z - uv[0]);
  r[1] = w * (y / z - uv[1]);
  /* d(uv)/d(p_FinCi) = [1/z 0 -x/z^2; 0 1/z -y/z^2], chained with R */
  const double a0 = 1.0 / z, a2 = -x / (z * z), b2 = -y / (z * z);
  for (int k = 0; k < 3; ++k) {
    Jp[k] = w * (a0 * R[k] + a2 * R[6 + k]);
    Jp[3 + k] = w * (a0 * R[3 + k] + b2 * R[6 + k]);
  }
}

static void plane_block(const double *p, const double *cp, double sigma_c, double *r, double Jp[3], double Jc[3]) {
  /* ceres/Factor_PointOnPlane.cpp:41-71 */
  const double d = norm3(cp);
  const double n[3] = {cp[0] / d, cp[1] / d, cp[2] / d};
  const double w = 1.0 / sigma_c;
  const double np = dot3(n, p);
  *r = -1.0 * w * (0.0 - (np - d)); /* :53 */
  for (int k = 0; k < 3; ++k) {
    Jp[k] = w * n[k];                                    /* :61 */
    Jc[k] = w * 1.0 / d * (p[k] - np * n[k] - d * n[k]); /* :67-68 */
  }
}

/* Cauchy loss rho(s) = log(1 + s) (ceres::CauchyLoss(1.0)); corrector (ceres corrector.cc): rho'' < 0 always, so residual and
 * Jacobian are scaled by sqrt(rho') */
static double cauchy_rho(double s) { return log(1.0 + s); }
static double cauchy_sqrt_rho1(double s) { return sqrt(1.0 / (1.0 + s)); }

/* Walks the residual blocks.  If J (dense, nrows x npar, row-major) is non-NULL it receives the corrected Jacobian and res the
 * corrected residuals; returns the cost 0.5 sum rho(s). */
static double popt_eval(const popt_t *o, const double *x, double *res, double *J) {
  const ovo_planeopt_problem *pr = o->pr;
  double cost = 0.0;
  int row = 0;
  const double *cp = (o->cp_off >= 0) ? x + o->cp_off : pr->cp;
  for (int f = 0; f < pr->n_feats; ++f) {
    const int sl = o->slot[f];
    const double *p = (sl >= 0) ? x + 3 * sl : pr->p_FinG + 3 * f;
    const int m = pr->n_obs[f];
    /* emits a constraint block */
#define OVO_CONSTRAINT(sigma)                                            \
  do {                                                                   \
    double r_, Jp_[3], Jc_[3];                                           \
    plane_block(p, cp, (sigma), &r_, Jp_, Jc_);                          \
    const double s_ = r_ * r_;                                           \
    cost += 0.5 * cauchy_rho(s_);                                        \
    if (sl >= 0 || o->cp_off >= 0) {                                     \
      const double a_ = cauchy_sqrt_rho1(s_);                            \
      if (res) res[row] = a_ * r_;                                       \
      if (J) {                                                           \
        double *Jr = J + (size_t)row * o->npar;                          \
        if (sl >= 0)                                                     \
          for (int k = 0; k < 3; ++k) Jr[3 * sl + k] = a_ * Jp_[k];      \
        if (o->cp_off >= 0)                                              \
          for (int k = 0; k < 3; ++k) Jr[o->cp_off + k] = a_ * Jc_[k];   \
      }                                                                  \
      ++row;                                                             \
    }                                                                    \
  } while (0)
    if (m == 0) OVO_CONSTRAINT(2.00 * pr->sigma_c); /* :276-279 slam_inflation */
    for (int k = 0; k < m; ++k) {
      const int ob = pr->obs_start[f] + k;
      double r2[2], Jp[6];
      reproj_block(pr->R_GtoC + 9 * ob, pr->p_CinG + 3 * ob, pr->uv_norm + 2 * ob, pr->sigma_px_norm, p, r2, Jp);
      const double s = r2[0] * r2[0] + r2[1] * r2[1];
      cost += 0.5 * cauchy_rho(s);
      if (sl >= 0) {
        const double a = cauchy_sqrt_rho1(s);
        if (res) {
          res[row] = a * r2[0];
          res[row + 1] = a * r2[1];
        }
        if (J)
          for (int q = 0; q < 3; ++q) {
            J[(size_t)row * o->npar + 3 * sl + q] = a * Jp[q];
            J[(size_t)(row + 1) * o->npar + 3 * sl + q] = a * Jp[3 + q];
          }
        row += 2;
      }
      OVO_CONSTRAINT(pr->sigma_c); /* :366-368 */
    }
#undef OVO_CONSTRAINT
  }
  return cost;
}

/* solves (J^T J + diag(D)^2) y = J^T r by Cholesky (what DENSE_SCHUR computes, without the elimination order); 0 on failure */
static int popt_solve(int m, int n, const double *J, const double *r, const double *D, double *y, double *work) {
  double *H = work; /* n x n */
  for (int i = 0; i < n; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = 0.0;
      for (int k = 0; k < m; ++k) s += J[(size_t)k * n + i] * J[(size_t)k * n + j];
      if (i == j) s += D[i] * D[i];
      H[i * n + j] = s;
    }
  for (int i = 0; i < n; ++i) {
    double s = 0.0;
    for (int k = 0; k < m; ++k) s += J[(size_t)k * n + i] * r[k];
    y[i] = s;
  }
  for (int j = 0; j < n; ++j) {
    double d = H[j * n + j];
    for (int k = 0; k < j; ++k) d -= H[j * n + k] * H[j * n + k];
    if (!(d > 0.0)) return 0;
    d = sqrt(d);
    H[j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double s = H[i * n + j];
      for (int k = 0; k < j; ++k) s -= H[i * n + k] * H[j * n + k];
      H[i * n + j] = s / d;
    }
  }
  for (int i = 0; i < n; ++i) {
    double s = y[i];
    for (int k = 0; k < i; ++k) s -= H[i * n + k] * y[k];
    y[i] = s / H[i * n + i];
  }
  for (int i = n - 1; i >= 0; --i) {
    double s = y[i];
    for (int k = i + 1; k < n; ++k) s -= H[k * n + i] * y[k];
    y[i] = s / H[i * n + i];
  }
  for (int i = 0; i < n; ++i)
    if (!isfinite(y[i])) return 0;
  return 1;
}

int ovo_optimize_plane(const ovo_planeopt_problem *pr, double cp_out[3], double *p_out, unsigned char *kept, int *n_kept,
                       int *iterations) {
  const double min_inlier_ratio = 0.80, max_error_threshold = 0.03; /* :207-211 */
  const int nf = pr->n_feats;
  const int ratio_n = (int)((double)nf * min_inlier_ratio);
  const size_t min_on_plane = (size_t)(4 > ratio_n ? 4 : ratio_n);
  for (int f = 0; f < nf; ++f) kept[f] = 0;
  *n_kept = 0;
  if (iterations) *iterations = 0;
  memcpy(cp_out, pr->cp, sizeof(double) * 3);
  memcpy(p_out, pr->p_FinG, sizeof(double) * 3 * (size_t)nf);
  if ((!pr->fix_plane && nf < 4) || (pr->fix_plane && nf == 0)) return 0; /* :214-217 */

  popt_t o;
  o.pr = pr;
  o.slot = (int *)malloc(sizeof(int) * (size_t)nf);
  o.nfree = 0;
  int rows = 0;
  for (int f = 0; f < nf; ++f) {
    o.slot[f] = (pr->n_obs[f] > 0) ? o.nfree++ : -1; /* :274-279: features without measurements are constant */
    if (pr->n_obs[f] > 0) rows += 3 * pr->n_obs[f];
    else if (!pr->fix_plane) rows += 1;
  }
  o.cp_off = pr->fix_plane ? -1 : 3 * o.nfree;
  o.npar = 3 * o.nfree + (pr->fix_plane ? 0 : 3);
  o.nres_rows = rows;
  const int n = o.npar, m = rows;

  int converged = 0;
  double *x = (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double));
  for (int f = 0; f < nf; ++f)
    if (o.slot[f] >= 0) memcpy(x + 3 * o.slot[f], pr->p_FinG + 3 * f, sizeof(double) * 3);
  if (o.cp_off >= 0) memcpy(x + o.cp_off, pr->cp, sizeof(double) * 3);

  if (n == 0) {
    converged = 1; /* Ceres: "Function tolerance reached. No non-constant parameter blocks found." */
  } else {
    /* ---- ceres::internal::TrustRegionMinimizer with DoglegStrategy (TRADITIONAL_DOGLEG), default options ---- */
    const int max_iter = 12;                 /* :385 */
    const double function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;
    const double min_relative_decrease = 1e-3, min_radius = 1e-32;
    double radius = 1e4, mu = 1e-8;
    const double min_mu = 1e-8, max_mu = 1.0, mu_inc = 10.0;
    int reuse = 0, invalid_run = 0;
    double *J = (double *)calloc((size_t)m * n, sizeof(double)), *res = (double *)calloc((size_t)m, sizeof(double));
    double *scale = (double *)malloc(sizeof(double) * n), *g = (double *)malloc(sizeof(double) * n);
    double *diag = (double *)malloc(sizeof(double) * n), *gn = (double *)malloc(sizeof(double) * n);
    double *step = (double *)malloc(sizeof(double) * n), *xc = (double *)malloc(sizeof(double) * n);
    double *D = (double *)malloc(sizeof(double) * n), *work = (double *)malloc(sizeof(double) * (size_t)n * n);
    double *tmp = (double *)malloc(sizeof(double) * (size_t)(m > n ? m : n));
    double alpha = 0.0, dogleg_norm = 0.0;

    double cost = popt_eval(&o, x, res, J);
    /* Jacobi scaling, estimated once at the start: 1 / (1 + |column|) */
    for (int jx = 0; jx < n; ++jx) {
      double s = 0.0;
      for (int k = 0; k < m; ++k) s += J[(size_t)k * n + jx] * J[(size_t)k * n + jx];
      scale[jx] = 1.0 / (1.0 + sqrt(s));
    }
#define OVO_SCALE_J()                 \
  for (int k_ = 0; k_ < m; ++k_)      \
    for (int j_ = 0; j_ < n; ++j_) J[(size_t)k_ * n + j_] *= scale[j_]
    /* unscaled gradient max-norm for the gradient tolerance */
    double gmax = 0.0;
    for (int jx = 0; jx < n; ++jx) {
      double s = 0.0;
      for (int k = 0; k < m; ++k) s += J[(size_t)k * n + jx] * res[k];
      if (fabs(s) > gmax) gmax = fabs(s);
    }
    OVO_SCALE_J();
    double xnorm = 0.0;
    for (int jx = 0; jx < n; ++jx) xnorm += x[jx] * x[jx];
    xnorm = sqrt(xnorm);
    int iter = 0;
    while (!converged) {
      /* FinalizeIterationAndCheckIfMinimizerCanContinue: iterations, gradient, radius - in this order */
      if (iter >= max_iter) break; /* NO_CONVERGENCE */
      if (gmax <= gradient_tolerance || radius <= min_radius) {
        converged = 1; /* "Minimum trust region radius reached" is CONVERGENCE as well */
        break;
      }
      ++iter;
      /* ---- DoglegStrategy::ComputeStep ---- */
      int step_ok = 1;
      if (!reuse) {
        reuse = 1; /* set by ComputeStep itself; StepAccepted / StepIsInvalid clear it */
        for (int jx = 0; jx < n; ++jx) {
          double s = 0.0, q = 0.0;
          for (int k = 0; k < m; ++k) {
            s += J[(size_t)k * n + jx] * res[k];
            q += J[(size_t)k * n + jx] * J[(size_t)k * n + jx];
          }
          g[jx] = s;
          if (q < 1e-6) q = 1e-6;
          if (q > 1e32) q = 1e32;
          diag[jx] = sqrt(q);
        }
        /* Cauchy point: alpha = |g/d|^2 / |J (g / d^2)|^2  (in the diagonally scaled space) */
        {
          double num = 0.0, den = 0.0;
          for (int jx = 0; jx < n; ++jx) num += (g[jx] / diag[jx]) * (g[jx] / diag[jx]);
          for (int k = 0; k < m; ++k) {
            double s = 0.0;
            for (int jx = 0; jx < n; ++jx) s += J[(size_t)k * n + jx] * (g[jx] / (diag[jx] * diag[jx]));
            den += s * s;
          }
          alpha = num / den;
        }
        /* Gauss-Newton step with the regulariser sqrt(mu) * diag; mu grows on failure */
        int solved = 0;
        while (mu < max_mu) {
          for (int jx = 0; jx < n; ++jx) D[jx] = diag[jx] * sqrt(mu);
          if (popt_solve(m, n, J, res, D, gn, work)) {
            solved = 1;
            break;
          }
          mu *= mu_inc;
        }
        if (!solved) step_ok = 0;
        else
          for (int jx = 0; jx < n; ++jx) gn[jx] *= -diag[jx]; /* scaled space */
      }
      double model_change = 0.0;
      if (step_ok) {
        /* traditional dogleg in the scaled space (gradient g / d) */
        double gn_norm = 0.0, gs_norm = 0.0;
        for (int jx = 0; jx < n; ++jx) {
          gn_norm += gn[jx] * gn[jx];
          gs_norm += (g[jx] / diag[jx]) * (g[jx] / diag[jx]);
        }
        gn_norm = sqrt(gn_norm);
        gs_norm = sqrt(gs_norm);
        if (gn_norm <= radius) {
          for (int jx = 0; jx < n; ++jx) step[jx] = gn[jx];
          dogleg_norm = gn_norm;
        } else if (gs_norm * alpha >= radius) {
          for (int jx = 0; jx < n; ++jx) step[jx] = -(radius / gs_norm) * (g[jx] / diag[jx]);
          dogleg_norm = radius;
        } else {
          double b_dot_a = 0.0;
          for (int jx = 0; jx < n; ++jx) b_dot_a += -alpha * (g[jx] / diag[jx]) * gn[jx];
          const double a2 = alpha * alpha * gs_norm * gs_norm;
          const double bma2 = a2 - 2.0 * b_dot_a + gn_norm * gn_norm;
          const double c = b_dot_a - a2;
          const double d = sqrt(c * c + bma2 * (radius * radius - a2));
          const double beta = (c <= 0) ? (d - c) / bma2 : (radius * radius - a2) / (d + c);
          for (int jx = 0; jx < n; ++jx) step[jx] = (-alpha * (1.0 - beta)) * (g[jx] / diag[jx]) + beta * gn[jx];
          double sn = 0.0;
          for (int jx = 0; jx < n; ++jx) sn += step[jx] * step[jx];
          dogleg_norm = sqrt(sn);
        }
        for (int jx = 0; jx < n; ++jx) step[jx] /= diag[jx];
        /* model_cost_change = -(J step) . (res + J step / 2) */
        for (int k = 0; k < m; ++k) {
          double s = 0.0;
          for (int jx = 0; jx < n; ++jx) s += J[(size_t)k * n + jx] * step[jx];
          model_change += -s * (res[k] + 0.5 * s);
        }
        if (!(model_change > 0.0)) step_ok = 0;
      }
      if (!step_ok) { /* HandleInvalidStep */
        if (++invalid_run >= 5) break;
        mu *= mu_inc;
        reuse = 0;
        continue;
      }
      invalid_run = 0;
      double snorm = 0.0;
      for (int jx = 0; jx < n; ++jx) {
        const double dlt = step[jx] * scale[jx];
        xc[jx] = x[jx] + dlt;
        snorm += dlt * dlt;
      }
      snorm = sqrt(snorm);
      const double cand = popt_eval(&o, xc, NULL, NULL);
      if (snorm <= parameter_tolerance * (xnorm + parameter_tolerance)) {
        converged = 1;
        break;
      }
      if (fabs(cost - cand) <= function_tolerance * cost) {
        converged = 1;
        break;
      }
      const double quality = (cost - cand) / model_change;
      if (getenv("OVO_PLANEOPT_TRACE"))
        fprintf(stderr, "it %d cost %.9e cand %.9e model %.3e q %.3f radius %.3e |step| %.3e mu %.1e reuse %d gmax %.2e\n", iter, cost,
                cand, model_change, quality, radius, snorm, mu, reuse, gmax);
      if (quality > min_relative_decrease) { /* successful step */
        memcpy(x, xc, sizeof(double) * n);
        memset(J, 0, sizeof(double) * (size_t)m * n);
        cost = popt_eval(&o, x, res, J);
        gmax = 0.0;
        for (int jx = 0; jx < n; ++jx) {
          double s = 0.0;
          for (int k = 0; k < m; ++k) s += J[(size_t)k * n + jx] * res[k];
          if (fabs(s) > gmax) gmax = fabs(s);
        }
        OVO_SCALE_J();
        xnorm = 0.0;
        for (int jx = 0; jx < n; ++jx) xnorm += x[jx] * x[jx];
        xnorm = sqrt(xnorm);
        if (quality < 0.25) radius *= 0.5; /* StepAccepted */
        if (quality > 0.75) radius = fmax(radius, 3.0 * dogleg_norm);
        mu = fmax(min_mu, 2.0 * mu / mu_inc);
        reuse = 0;
      } else { /* StepRejected */
        radius *= 0.5;
        reuse = 1;
      }
    }
#undef OVO_SCALE_J
    if (iterations) *iterations = iter;
    free(J);
    free(res);
    free(scale);
    free(g);
    free(diag);
    free(gn);
    free(step);
    free(xc);
    free(D);
    free(work);
    free(tmp);
  }

  int ok = 0;
  if (converged) { /* :422-429 otherwise */
    double cp[3], abcd[4];
    memcpy(cp, (o.cp_off >= 0) ? x + o.cp_off : pr->cp, sizeof(cp)); /* :432-441 */
    const double d = norm3(cp);
    abcd[0] = cp[0] / d;
    abcd[1] = cp[1] / d;
    abcd[2] = cp[2] / d;
    abcd[3] = -d;
    /* current camera pose from the IMU pose and the extrinsics, :444-453 */
    double R_GtoC[9], p_CinG[3];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        double s = 0.0;
        for (int k = 0; k < 3; ++k) s += pr->R_ItoC[3 * i + k] * pr->R_GtoI[3 * k + j];
        R_GtoC[3 * i + j] = s;
      }
    for (int i = 0; i < 3; ++i) {
      double s = 0.0;
      for (int k = 0; k < 3; ++k) s += R_GtoC[3 * k + i] * pr->p_IinC[k];
      p_CinG[i] = pr->p_IinG[i] - s;
    }
    int cnt = 0;
    for (int f = 0; f < nf; ++f) { /* :456-485 */
      const double *pb = pr->p_FinG + 3 * f;
      const double *pa = (o.slot[f] >= 0) ? x + 3 * o.slot[f] : pb;
      const double e = pt_plane_dist(pb, abcd); /* the OLD estimate against the NEW plane, as written at :467 */
      if (fabs(e) >= max_error_threshold) continue;
      if (isnan(norm3(pa))) continue;
      const double dd[3] = {pa[0] - p_CinG[0], pa[1] - p_CinG[1], pa[2] - p_CinG[2]};
      const double z = R_GtoC[6] * dd[0] + R_GtoC[7] * dd[1] + R_GtoC[8] * dd[2];
      if (z < 0.1) continue;
      kept[f] = 1;
      ++cnt;
    }
    /* :491-498 */
    const int fail = (nf != 1 && (size_t)cnt < min_on_plane) || (pr->fix_plane && nf == 1 && cnt == 0);
    if (!fail) {
      ok = 1;
      memcpy(cp_out, cp, sizeof(cp));
      for (int f = 0; f < nf; ++f)
        if (kept[f] && o.slot[f] >= 0) memcpy(p_out + 3 * f, x + 3 * o.slot[f], sizeof(double) * 3);
      *n_kept = cnt;
    } else {
      for (int f = 0; f < nf; ++f) kept[f] = 0;
    }
  }
  free(x);
  free(o.slot);
  return ok;
}

/* cost / gradient at a point, for the finite-difference and stationarity pins of tests/ */
double ovo_planeopt_cost(const ovo_planeopt_problem *pr, const double *p_FinG, const double *cp, double *grad_p, double *grad_cp) {
  ovo_planeopt_problem q = *pr;
  q.p_FinG = p_FinG;
  memcpy(q.cp, cp, sizeof(double) * 3);
  q.fix_plane = 0;
  popt_t o;
  o.pr = &q;
  const int nf = q.n_feats;
  o.slot = (int *)malloc(sizeof(int) * (size_t)nf);
  o.nfree = 0;
  int rows = 0;
  for (int f = 0; f < nf; ++f) {
    o.slot[f] = (q.n_obs[f] > 0) ? o.nfree++ : -1;
    rows += (q.n_obs[f] > 0) ? 3 * q.n_obs[f] : 1;
  }
  o.cp_off = 3 * o.nfree;
  o.npar = 3 * o.nfree + 3;
  o.nres_rows = rows;
  double *x = (double *)calloc((size_t)o.npar, sizeof(double));
  for (int f = 0; f < nf; ++f)
    if (o.slot[f] >= 0) memcpy(x + 3 * o.slot[f], p_FinG + 3 * f, sizeof(double) * 3);
  memcpy(x + o.cp_off, cp, sizeof(double) * 3);
  double *J = (double *)calloc((size_t)rows * o.npar, sizeof(double)), *res = (double *)calloc((size_t)rows, sizeof(double));
  const double cost = popt_eval(&o, x, res, J);
  if (grad_p) memset(grad_p, 0, sizeof(double) * 3 * (size_t)nf);
  for (int jx = 0; jx < o.npar; ++jx) {
    double s = 0.0;
    for (int k = 0; k < rows; ++k) s += J[(size_t)k * o.npar + jx] * res[k];
    if (jx >= o.cp_off) {
      if (grad_cp) grad_cp[jx - o.cp_off] = s;
    } else if (grad_p) {
      for (int f = 0; f < nf; ++f)
        if (o.slot[f] == jx / 3) grad_p[3 * f + jx % 3] = s;
    }
  }
  free(J);
  free(res);
  free(x);
  free(o.slot);
  return cost;
}
