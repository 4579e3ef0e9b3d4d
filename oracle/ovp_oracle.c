/*
 * TEST INFRASTRUCTURE ONLY - see ovp_oracle.h.  Plain C99 restatement of the reference CPU path, in the
 * reference's loop order.  Citations are relative to /root/reference/ov_plane/src/.
 */
#include "ovp_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define CM(A, ld, r, c) ((A)[(size_t)(c) * (size_t)(ld) + (size_t)(r)])

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ---------------------------------------------------------------------------------------------
 * ext ov_core quat_ops.h (JPL), SURVEY.md Appendix A
 * ------------------------------------------------------------------------------------------- */
void ovo_quat_2_rot(const double q[4], double R[9]) {
  /* R = (2 q4^2 - 1) I - 2 q4 [qv]x + 2 qv qv^T */
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double a = 2.0 * w * w - 1.0;
  R[0] = a + 2.0 * x * x;
  R[1] = 2.0 * w * z + 2.0 * x * y;
  R[2] = -2.0 * w * y + 2.0 * x * z;
  R[3] = -2.0 * w * z + 2.0 * y * x;
  R[4] = a + 2.0 * y * y;
  R[5] = 2.0 * w * x + 2.0 * y * z;
  R[6] = 2.0 * w * y + 2.0 * z * x;
  R[7] = -2.0 * w * x + 2.0 * z * y;
  R[8] = a + 2.0 * z * z;
}

static void quat_multiply(const double q[4], const double p[4], double out[4]) {
  /* [[q4 I - [qv]x, qv], [-qv^T, q4]] * p, then q4>=0 and unit norm */
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  double r[4];
  r[0] = w * p[0] + z * p[1] - y * p[2] + x * p[3];
  r[1] = -z * p[0] + w * p[1] + x * p[2] + y * p[3];
  r[2] = y * p[0] - x * p[1] + w * p[2] + z * p[3];
  r[3] = -x * p[0] - y * p[1] - z * p[2] + w * p[3];
  if (r[3] < 0) {
    r[0] = -r[0];
    r[1] = -r[1];
    r[2] = -r[2];
    r[3] = -r[3];
  }
  const double n = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3]);
  out[0] = r[0] / n;
  out[1] = r[1] / n;
  out[2] = r[2] / n;
  out[3] = r[3] / n;
}

void ovo_quat_update(double q[4], const double dth[3]) {
  /* ext JPLQuat::update: dq = quatnorm([0.5 dth; 1]); q <- dq (x) q */
  double dq[4] = {0.5 * dth[0], 0.5 * dth[1], 0.5 * dth[2], 1.0};
  const double n = sqrt(dq[0] * dq[0] + dq[1] * dq[1] + dq[2] * dq[2] + dq[3] * dq[3]);
  dq[0] /= n;
  dq[1] /= n;
  dq[2] /= n;
  dq[3] /= n;
  double out[4];
  quat_multiply(dq, q, out);
  memcpy(q, out, sizeof(out));
}

/* ---------------------------------------------------------------------------------------------
 * ext ov_core CamRadtan::distort_d / compute_distort_jacobian (call sites update/UpdaterHelper.cpp:365,389)
 * ------------------------------------------------------------------------------------------- */
void ovo_radtan_distort(const double v[8], const double uvn[2], double uvd[2]) {
  const double x = uvn[0], y = uvn[1];
  const double r2 = x * x + y * y, r4 = r2 * r2;
  const double x1 = x * (1 + v[4] * r2 + v[5] * r4) + 2 * v[6] * x * y + v[7] * (r2 + 2 * x * x);
  const double y1 = y * (1 + v[4] * r2 + v[5] * r4) + v[6] * (r2 + 2 * y * y) + 2 * v[7] * x * y;
  uvd[0] = v[0] * x1 + v[2];
  uvd[1] = v[1] * y1 + v[3];
}

void ovo_radtan_jacobian(const double v[8], const double uvn[2], double dz_dzn[4], double dz_dzeta[16]) {
  /* dz_dzn row-major 2x2, dz_dzeta row-major 2x8 */
  const double x = uvn[0], y = uvn[1];
  const double r2 = x * x + y * y, r4 = r2 * r2;
  const double fx = v[0], fy = v[1], k1 = v[4], k2 = v[5], p1 = v[6], p2 = v[7];
  const double g = 1 + k1 * r2 + k2 * r4;
  const double x1 = x * g + 2 * p1 * x * y + p2 * (r2 + 2 * x * x);
  const double y1 = y * g + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y;
  dz_dzn[0] = fx * (g + 2 * k1 * x * x + 4 * k2 * x * x * r2 + 2 * p1 * y + 6 * p2 * x);
  dz_dzn[1] = fx * (2 * k1 * x * y + 4 * k2 * x * y * r2 + 2 * p1 * x + 2 * p2 * y);
  dz_dzn[2] = fy * (2 * k1 * x * y + 4 * k2 * x * y * r2 + 2 * p1 * x + 2 * p2 * y);
  dz_dzn[3] = fy * (g + 2 * k1 * y * y + 4 * k2 * y * y * r2 + 6 * p1 * y + 2 * p2 * x);
  memset(dz_dzeta, 0, 16 * sizeof(double));
  dz_dzeta[0] = x1;
  dz_dzeta[2] = 1;
  dz_dzeta[4] = fx * x * r2;
  dz_dzeta[5] = fx * x * r4;
  dz_dzeta[6] = 2 * fx * x * y;
  dz_dzeta[7] = fx * (r2 + 2 * x * x);
  dz_dzeta[8 + 1] = y1;
  dz_dzeta[8 + 3] = 1;
  dz_dzeta[8 + 4] = fy * y * r2;
  dz_dzeta[8 + 5] = fy * y * r4;
  dz_dzeta[8 + 6] = fy * (r2 + 2 * y * y);
  dz_dzeta[8 + 7] = 2 * fy * x * y;
}

/* ---------------------------------------------------------------------------------------------
 * Eigen::JacobiRotation<double>::makeGivens (real case) and applyOnTheLeft(0,1,G.adjoint())
 * ------------------------------------------------------------------------------------------- */
void ovo_make_givens(double p, double q, double *c, double *s) {
  if (q == 0.0) {
    *c = p < 0 ? -1.0 : 1.0;
    *s = 0.0;
  } else if (p == 0.0) {
    *c = 0.0;
    *s = q < 0 ? 1.0 : -1.0;
  } else if (fabs(p) > fabs(q)) {
    double t = q / p;
    double u = sqrt(1.0 + t * t);
    if (p < 0) u = -u;
    *c = 1.0 / u;
    *s = -t * (*c);
  } else {
    double t = p / q;
    double u = sqrt(1.0 + t * t);
    if (q < 0) u = -u;
    *s = -1.0 / u;
    *c = -t * (*s);
  }
}

/* rows (r0, r0+1) of a column-major matrix, columns [c0, c1):  x' = c x - s y ; y' = s x + c y */
static inline void rot_rows(double *A, int ld, int r0, int c0, int c1, double c, double s) {
  for (int j = c0; j < c1; ++j) {
    double *col = A + (size_t)j * (size_t)ld;
    const double x = col[r0], y = col[r0 + 1];
    col[r0] = c * x - s * y;
    col[r0 + 1] = s * x + c * y;
  }
}

/* ---------------------------------------------------------------------------------------------
 * chi-square 0.95 quantile  (boost::math::quantile(chi_squared(k), 0.95), update/UpdaterMSCKF.cpp:59-62,749-750)
 * regularised lower incomplete gamma by series / continued fraction + safeguarded Newton.
 * ------------------------------------------------------------------------------------------- */
static double gammap(double a, double x) {
  if (x <= 0) return 0.0;
  const double gln = lgamma(a);
  if (x < a + 1.0) {
    double ap = a, sum = 1.0 / a, del = sum;
    for (int n = 0; n < 100000; ++n) {
      ap += 1.0;
      del *= x / ap;
      sum += del;
      if (fabs(del) < fabs(sum) * 1e-17) break;
    }
    return sum * exp(-x + a * log(x) - gln);
  }
  double b = x + 1.0 - a, c = 1.0 / 1e-300, d = 1.0 / b, h = d;
  for (int i = 1; i < 100000; ++i) {
    const double an = -i * (i - a);
    b += 2.0;
    d = an * d + b;
    if (fabs(d) < 1e-300) d = 1e-300;
    c = b + an / c;
    if (fabs(c) < 1e-300) c = 1e-300;
    d = 1.0 / d;
    const double del = d * c;
    h *= del;
    if (fabs(del - 1.0) < 1e-17) break;
  }
  return 1.0 - exp(-x + a * log(x) - gln) * h;
}

double ovo_chi2_quantile_095(int dof) {
  const double p = 0.95, a = 0.5 * (double)dof;
  /* Wilson-Hilferty start */
  const double z = 1.6448536269514722;
  double t = 1.0 - 2.0 / (9.0 * dof) + z * sqrt(2.0 / (9.0 * dof));
  double x = 0.5 * dof * t * t * t;
  if (x <= 0) x = 0.5;
  double lo = 0.0, hi = 1e300;
  for (int it = 0; it < 200; ++it) {
    const double f = gammap(a, x) - p;
    if (f > 0) hi = x; else lo = x;
    const double dens = exp((a - 1.0) * log(x) - x - lgamma(a));
    double xn = x - f / dens;
    if (!(xn > lo && xn < hi)) xn = (hi < 1e299) ? 0.5 * (lo + hi) : 2.0 * x;
    if (fabs(xn - x) <= 1e-15 * fabs(xn)) {
      x = xn;
      break;
    }
    x = xn;
  }
  return 2.0 * x;
}

/* ---------------------------------------------------------------------------------------------
 * Eigen LLT (lower), column-major in place
 * ------------------------------------------------------------------------------------------- */
int ovo_llt(double *A, int n, int ld) {
  for (int j = 0; j < n; ++j) {
    double d = CM(A, ld, j, j);
    for (int k = 0; k < j; ++k) d -= CM(A, ld, j, k) * CM(A, ld, j, k);
    if (!(d > 0.0)) return j + 1;
    d = sqrt(d);
    CM(A, ld, j, j) = d;
    for (int i = j + 1; i < n; ++i) {
      double s = CM(A, ld, i, j);
      for (int k = 0; k < j; ++k) s -= CM(A, ld, i, k) * CM(A, ld, j, k);
      CM(A, ld, i, j) = s / d;
    }
  }
  return 0;
}

/* solve L L^T x = b in place, L lower col-major */
static void llt_solve_vec(const double *L, int n, int ld, double *b) {
  for (int i = 0; i < n; ++i) {
    double s = b[i];
    for (int k = 0; k < i; ++k) s -= CM(L, ld, i, k) * b[k];
    b[i] = s / CM(L, ld, i, i);
  }
  for (int i = n - 1; i >= 0; --i) {
    double s = b[i];
    for (int k = i + 1; k < n; ++k) s -= CM(L, ld, k, i) * b[k];
    b[i] = s / CM(L, ld, i, i);
  }
}

/* ---------------------------------------------------------------------------------------------
 * ext ov_core CamEqui::distort_d / compute_distort_jacobian (fisheye; same call sites), value = fx fy cx cy k1 k2 k3 k4:
 *   theta = atan(r), theta_d = theta + k1 theta^3 + k2 theta^5 + k3 theta^7 + k4 theta^9, uv = f * xy * theta_d / r + c
 * ------------------------------------------------------------------------------------------- */
void ovo_equi_distort(const double v[8], const double uvn[2], double uvd[2]) {
  const double r = sqrt(uvn[0] * uvn[0] + uvn[1] * uvn[1]);
  const double th = atan(r);
  const double th_d = th + v[4] * pow(th, 3) + v[5] * pow(th, 5) + v[6] * pow(th, 7) + v[7] * pow(th, 9);
  const double inv_r = (r > 1e-8) ? 1.0 / r : 1.0;
  const double cdist = (r > 1e-8) ? th_d * inv_r : 1.0;
  uvd[0] = v[0] * (uvn[0] * cdist) + v[2];
  uvd[1] = v[1] * (uvn[1] * cdist) + v[3];
}

void ovo_equi_jacobian(const double v[8], const double uvn[2], double dz_dzn[4], double dz_dzeta[16]) {
  const double x = uvn[0], y = uvn[1];
  const double r = sqrt(x * x + y * y);
  const double th = atan(r);
  const double th_d = th + v[4] * pow(th, 3) + v[5] * pow(th, 5) + v[6] * pow(th, 7) + v[7] * pow(th, 9);
  const double inv_r = (r > 1e-8) ? 1.0 / r : 1.0;
  const double cdist = (r > 1e-8) ? th_d * inv_r : 1.0;
  /* duv_dxy (dxy_dxyn + (dxy_dr + dxy_dthd dthd_dth dth_dr) dr_dxyn) */
  const double dthd_dth = 1 + 3 * v[4] * pow(th, 2) + 5 * v[5] * pow(th, 4) + 7 * v[6] * pow(th, 6) + 9 * v[7] * pow(th, 8);
  const double dth_dr = 1 / (r * r + 1);
  const double dxy_dr[2] = {-x * th_d * inv_r * inv_r, -y * th_d * inv_r * inv_r};
  const double dxy_dthd[2] = {x * inv_r, y * inv_r};
  const double dr_dxyn[2] = {x * inv_r, y * inv_r};
  double M[4];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      M[2 * i + j] = ((i == j) ? th_d * inv_r : 0.0) + (dxy_dr[i] + dxy_dthd[i] * dthd_dth * dth_dr) * dr_dxyn[j];
  dz_dzn[0] = v[0] * M[0];
  dz_dzn[1] = v[0] * M[1];
  dz_dzn[2] = v[1] * M[2];
  dz_dzn[3] = v[1] * M[3];
  memset(dz_dzeta, 0, 16 * sizeof(double));
  dz_dzeta[0] = x * cdist;
  dz_dzeta[2] = 1;
  dz_dzeta[8 + 1] = y * cdist;
  dz_dzeta[8 + 3] = 1;
  for (int k = 0; k < 4; ++k) {
    const double pw = pow(th, 3 + 2 * k);
    dz_dzeta[4 + k] = v[0] * x * inv_r * pw;
    dz_dzeta[8 + 4 + k] = v[1] * y * inv_r * pw;
  }
}

/* ---------------------------------------------------------------------------------------------
 * small 3x3 helpers (row-major)
 * ------------------------------------------------------------------------------------------- */
static void mat3_mul(const double *A, const double *B, double *C) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
static void mat3_vec(const double *A, const double *v, double *o) {
  for (int i = 0; i < 3; ++i) o[i] = A[3 * i] * v[0] + A[3 * i + 1] * v[1] + A[3 * i + 2] * v[2];
}
static void skew3(const double *w, double *S) {
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

/* ---------------------------------------------------------------------------------------------
 * update/UpdaterHelper.cpp:195-513  (GLOBAL_3D: dpfg_dlambda = I, :39-43)
 * ------------------------------------------------------------------------------------------- */
int ovo_feature_jacobian_full(const ovo_opts *o, const ovo_state *st, const ovo_feats *fb, int f, double sigma_c,
                              int planeid, const double *cp, const double *cp_fej, int plane_state_id, double *H_f,
                              double *H_x, double *res, int *rows_out, int *cols_out, int *hf_cols_out,
                              int *order_id, int *order_size, int *n_order_out) {
  const int m = fb->n_meas[f];
  const int *cidx = fb->clone_idx + (size_t)f * fb->max_meas;
  const float *uv = fb->uv + (size_t)f * fb->max_meas * 2;
  const double *p_FinG = fb->p_FinG + (size_t)f * 3;
  /* MSCKF features: fej == value (UpdaterMSCKF.cpp:499-500,721-722); SLAM landmarks carry their first estimate */
  const double *p_FinG_fej = fb->p_FinG_fej ? fb->p_FinG_fej + (size_t)f * 3 : p_FinG;

  /* column bookkeeping :205-277 */
  int n_order = 0, total_hx = 0;
  int col_calib = -1, col_intr = -1, col_plane = -1;
  int *col_clone = (int *)malloc(sizeof(int) * (size_t)(st->n_clones > 0 ? st->n_clones : 1));
  for (int i = 0; i < st->n_clones; ++i) col_clone[i] = -1;
  if (o->do_calib_camera_pose) {
    col_calib = total_hx;
    order_id[n_order] = st->calib_id;
    order_size[n_order++] = 6;
    total_hx += 6;
  }
  if (o->do_calib_camera_intrinsics) {
    col_intr = total_hx;
    order_id[n_order] = st->intr_id;
    order_size[n_order++] = 8;
    total_hx += 8;
  }
  for (int k = 0; k < m; ++k) {
    const int ci = cidx[k];
    if (col_clone[ci] < 0) {
      col_clone[ci] = total_hx;
      order_id[n_order] = st->clone_id[ci];
      order_size[n_order++] = 6;
      total_hx += 6;
    }
  }
  const int plane_in_state = plane_state_id >= 0;
  if (planeid != 0 && plane_in_state) {
    col_plane = total_hx;
    order_id[n_order] = plane_state_id;
    order_size[n_order++] = 3;
    total_hx += 3;
  }

  /* sizes :309-318 */
  int jacobsize = 3 + ((planeid != 0 && !plane_in_state) ? 3 : 0);
  int meassize = (planeid != 0) ? 3 * m : 2 * m;
  if (m == 0 && planeid != 0) meassize = 1;
  const int ld = meassize;
  memset(res, 0, sizeof(double) * (size_t)meassize);
  memset(H_f, 0, sizeof(double) * (size_t)meassize * (size_t)jacobsize);
  memset(H_x, 0, sizeof(double) * (size_t)meassize * (size_t)total_hx);

  const double white_px = 1.0 / o->sigma_px;
  double R_ItoC[9];
  ovo_quat_2_rot(st->calib_q, R_ItoC);
  const double *p_IinC = st->calib_p;

  int c = 0;
  for (int k = 0; k < m; ++k) {
    const int ci = cidx[k];
    double R_GtoIi[9];
    ovo_quat_2_rot(st->clone_q + 4 * ci, R_GtoIi);
    const double *p_IiinG = st->clone_p + 3 * ci;
    /* :356-370 */
    double d[3] = {p_FinG[0] - p_IiinG[0], p_FinG[1] - p_IiinG[1], p_FinG[2] - p_IiinG[2]};
    double p_FinIi[3], p_FinCi[3];
    mat3_vec(R_GtoIi, d, p_FinIi);
    mat3_vec(R_ItoC, p_FinIi, p_FinCi);
    p_FinCi[0] += p_IinC[0];
    p_FinCi[1] += p_IinC[1];
    p_FinCi[2] += p_IinC[2];
    double uv_norm[2] = {p_FinCi[0] / p_FinCi[2], p_FinCi[1] / p_FinCi[2]};
    double uv_dist[2];
    if (st->cam_fisheye) ovo_equi_distort(st->intrinsics, uv_norm, uv_dist);
    else ovo_radtan_distort(st->intrinsics, uv_norm, uv_dist);
    const double uv_m[2] = {(double)uv[2 * k], (double)uv[2 * k + 1]};
    res[c] = white_px * (uv_m[0] - uv_dist[0]);
    res[c + 1] = white_px * (uv_m[1] - uv_dist[1]);
    /* :376-385 FEJ */
    if (o->do_fej) {
      ovo_quat_2_rot(st->clone_q_fej + 4 * ci, R_GtoIi);
      p_IiinG = st->clone_p_fej + 3 * ci;
      d[0] = p_FinG_fej[0] - p_IiinG[0];
      d[1] = p_FinG_fej[1] - p_IiinG[1];
      d[2] = p_FinG_fej[2] - p_IiinG[2];
      mat3_vec(R_GtoIi, d, p_FinIi);
      mat3_vec(R_ItoC, p_FinIi, p_FinCi);
      p_FinCi[0] += p_IinC[0];
      p_FinCi[1] += p_IinC[1];
      p_FinCi[2] += p_IinC[2];
    }
    /* :388-401 */
    double dz_dzn[4], dz_dzeta[16];
    if (st->cam_fisheye) ovo_equi_jacobian(st->intrinsics, uv_norm, dz_dzn, dz_dzeta);
    else ovo_radtan_jacobian(st->intrinsics, uv_norm, dz_dzn, dz_dzeta);
    const double z = p_FinCi[2];
    const double dzn_dpfc[6] = {1 / z, 0, -p_FinCi[0] / (z * z), 0, 1 / z, -p_FinCi[1] / (z * z)};
    double dpfc_dpfg[9];
    mat3_mul(R_ItoC, R_GtoIi, dpfc_dpfg);
    double sk[9], Rsk[9];
    skew3(p_FinIi, sk);
    mat3_mul(R_ItoC, sk, Rsk);
    double dpfc_dclone[18]; /* 3x6 row-major */
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        dpfc_dclone[6 * i + j] = Rsk[3 * i + j];
        dpfc_dclone[6 * i + 3 + j] = -dpfc_dpfg[3 * i + j];
      }
    /* :407-408 */
    double dz_dpfc[6], dz_dpfg[6];
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 3; ++j) dz_dpfc[3 * i + j] = dz_dzn[2 * i] * dzn_dpfc[j] + dz_dzn[2 * i + 1] * dzn_dpfc[3 + j];
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 3; ++j)
        dz_dpfg[3 * i + j] =
            dz_dpfc[3 * i] * dpfc_dpfg[j] + dz_dpfc[3 * i + 1] * dpfc_dpfg[3 + j] + dz_dpfc[3 * i + 2] * dpfc_dpfg[6 + j];
    /* :411 */
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 3; ++j) CM(H_f, ld, c + i, j) = white_px * dz_dpfg[3 * i + j];
    /* :414 */
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 6; ++j)
        CM(H_x, ld, c + i, col_clone[ci] + j) =
            white_px * (dz_dpfc[3 * i] * dpfc_dclone[j] + dz_dpfc[3 * i + 1] * dpfc_dclone[6 + j] +
                        dz_dpfc[3 * i + 2] * dpfc_dclone[12 + j]);
    /* :426-435 */
    if (o->do_calib_camera_pose) {
      const double v[3] = {p_FinCi[0] - p_IinC[0], p_FinCi[1] - p_IinC[1], p_FinCi[2] - p_IinC[2]};
      double skc[9];
      skew3(v, skc);
      double dpfc_dcalib[18];
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
          dpfc_dcalib[6 * i + j] = skc[3 * i + j];
          dpfc_dcalib[6 * i + 3 + j] = (i == j) ? 1.0 : 0.0;
        }
      for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 6; ++j)
          CM(H_x, ld, c + i, col_calib + j) +=
              white_px * (dz_dpfc[3 * i] * dpfc_dcalib[j] + dz_dpfc[3 * i + 1] * dpfc_dcalib[6 + j] +
                          dz_dpfc[3 * i + 2] * dpfc_dcalib[12 + j]);
    }
    /* :438-440 */
    if (o->do_calib_camera_intrinsics) {
      for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 8; ++j) CM(H_x, ld, c + i, col_intr + j) = white_px * dz_dzeta[8 * i + j];
    }
    c += 2;
  }

  /* :448-512 point-on-plane constraint rows, one per measurement */
  if (planeid != 0) {
    const double white_c = 1.0 / sigma_c;
    const int reps = (m == 0) ? 1 : m;
    for (int rep = 0; rep < reps; ++rep) {
      double dd = sqrt(cp[0] * cp[0] + cp[1] * cp[1] + cp[2] * cp[2]);
      double n[3] = {cp[0] / dd, cp[1] / dd, cp[2] / dd};
      res[c] = white_c * (0.0 - (n[0] * p_FinG[0] + n[1] * p_FinG[1] + n[2] * p_FinG[2] - dd));
      const double *lp = p_FinG;
      if (o->do_fej) {
        lp = p_FinG_fej;
        dd = sqrt(cp_fej[0] * cp_fej[0] + cp_fej[1] * cp_fej[1] + cp_fej[2] * cp_fej[2]);
        n[0] = cp_fej[0] / dd;
        n[1] = cp_fej[1] / dd;
        n[2] = cp_fej[2] / dd;
      }
      const double ndp = n[0] * lp[0] + n[1] * lp[1] + n[2] * lp[2];
      for (int j = 0; j < 3; ++j) {
        const double hcp = white_c * 1.0 / dd * (lp[j] - ndp * n[j] - dd * n[j]);
        if (plane_in_state)
          CM(H_x, ld, c, col_plane + j) = hcp;
        else
          CM(H_f, ld, c, jacobsize - 3 + j) = hcp;
        CM(H_f, ld, c, j) = white_c * n[j];
      }
      c += 1;
    }
  }
  free(col_clone);
  *rows_out = meassize;
  *cols_out = total_hx;
  *hf_cols_out = jacobsize;
  *n_order_out = n_order;
  return 0;
}

/* ---------------------------------------------------------------------------------------------
 * update/UpdaterHelper.cpp:35-193  get_feature_jacobian_representation for the anchored / inverse-depth landmark
 * parameterisations, and their use in get_feature_jacobian_full (:296-302, :323-327, :411-421).
 *   rep: 0 GLOBAL_3D, 1 GLOBAL_FULL_INVERSE_DEPTH, 2 ANCHORED_3D, 3 ANCHORED_FULL_INVERSE_DEPTH,
 *        4 ANCHORED_MSCKF_INVERSE_DEPTH, 5 ANCHORED_INVERSE_DEPTH_SINGLE (ext ov_type::LandmarkRepresentation)
 * Inputs: p_FinG (value; MSCKF features have fej == value), the anchor clone slot.  Outputs (row-major): dlam [3 x nl]
 * (nl = 3, or 1 for rep 5), H_anc [3x6], H_cal [3x6] (the latter two zero for the global representations).
 * ------------------------------------------------------------------------------------------- */
static void inv_depth_jac(const double p[3], double J[9]) { /* d p / d (theta, phi, rho), :50-72 */
  const double rho = 1 / sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
  const double phi = acos(rho * p[2]);
  const double th = atan2(p[1], p[0]);
  const double st = sin(th), ct = cos(th), sp = sin(phi), cph = cos(phi);
  J[0] = -(1.0 / rho) * st * sp;
  J[1] = (1.0 / rho) * ct * cph;
  J[2] = -(1.0 / (rho * rho)) * ct * sp;
  J[3] = (1.0 / rho) * ct * sp;
  J[4] = (1.0 / rho) * st * cph;
  J[5] = -(1.0 / (rho * rho)) * st * sp;
  J[6] = 0.0;
  J[7] = -(1.0 / rho) * sp;
  J[8] = -(1.0 / (rho * rho)) * cph;
}

int ovo_feature_jacobian_representation(const ovo_opts *o, const ovo_state *st, int rep, const double p_FinG[3], int anchor_ci,
                                        double *dlam, int *nl_out, double H_anc[18], double H_cal[18]) {
  return ovo_feature_jacobian_representation_fej(o, st, rep, p_FinG, NULL, anchor_ci, dlam, nl_out, H_anc, H_cal);
}

/* p_FinG_fej: first estimate of a GLOBAL_FULL_INVERSE_DEPTH landmark (NULL: equal to the value, as for MSCKF features); the
 * anchored representations re-express the CURRENT global point in the first-estimate anchor frame instead (:91-98) */
int ovo_feature_jacobian_representation_fej(const ovo_opts *o, const ovo_state *st, int rep, const double p_FinG[3],
                                            const double *p_FinG_fej, int anchor_ci, double *dlam, int *nl_out, double H_anc[18],
                                            double H_cal[18]) {
  memset(H_anc, 0, 18 * sizeof(double));
  memset(H_cal, 0, 18 * sizeof(double));
  *nl_out = 3;
  if (rep == 0) { /* :39-43 */
    for (int i = 0; i < 9; ++i) dlam[i] = (i % 4 == 0) ? 1.0 : 0.0;
    return 0;
  }
  if (rep == 1) { /* :46-76 */
    inv_depth_jac((o->do_fej && p_FinG_fej) ? p_FinG_fej : p_FinG, dlam);
    return 0;
  }
  /* :83-101 anchor pose and calibration; with FEJ the anchor pose is the first estimate and p_FinA is re-expressed in it */
  double R_ItoC[9], R_GtoI[9], tmp[3], d[3], p_FinA[3];
  ovo_quat_2_rot(st->calib_q, R_ItoC);
  const double *p_IinC = st->calib_p;
  const double *aq = o->do_fej ? st->clone_q_fej : st->clone_q;
  const double *ap = o->do_fej ? st->clone_p_fej : st->clone_p;
  ovo_quat_2_rot(aq + 4 * anchor_ci, R_GtoI);
  const double *p_IinG = ap + 3 * anchor_ci;
  /* p_FinG_best == p_FinG (:93); into the (first-estimate) anchor frame (:95-97) */
  for (int k = 0; k < 3; ++k) d[k] = p_FinG[k] - p_IinG[k];
  mat3_vec(R_GtoI, d, tmp);
  mat3_vec(R_ItoC, tmp, p_FinA);
  for (int k = 0; k < 3; ++k) p_FinA[k] += p_IinC[k];
  double R_CtoG[9]; /* R_GtoI^T R_ItoC^T */
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double a = 0;
      for (int k = 0; k < 3; ++k) a += R_GtoI[3 * k + i] * R_ItoC[3 * j + k];
      R_CtoG[3 * i + j] = a;
    }
  /* :102-106 H_anc = [ -R_GtoI^T skew(R_ItoC^T (p_FinA - p_IinC)) , I ] */
  double q[3] = {p_FinA[0] - p_IinC[0], p_FinA[1] - p_IinC[1], p_FinA[2] - p_IinC[2]}, v[3], S[9];
  for (int i = 0; i < 3; ++i) v[i] = R_ItoC[i] * q[0] + R_ItoC[3 + i] * q[1] + R_ItoC[6 + i] * q[2];
  skew3(v, S);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double a = 0;
      for (int k = 0; k < 3; ++k) a += R_GtoI[3 * k + i] * S[3 * k + j];
      H_anc[6 * i + j] = -a;
      H_anc[6 * i + 3 + j] = (i == j) ? 1.0 : 0.0;
    }
  if (o->do_calib_camera_pose) { /* :113-119 H_calib = [ -R_CtoG skew(p_FinA - p_IinC) , -R_CtoG ] */
    skew3(q, S);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        double a = 0;
        for (int k = 0; k < 3; ++k) a += R_CtoG[3 * i + k] * S[3 * k + j];
        H_cal[6 * i + j] = -a;
        H_cal[6 * i + 3 + j] = -R_CtoG[3 * i + j];
      }
  }
  double J[9];
  if (rep == 2) { /* :122-125 */
    memcpy(dlam, R_CtoG, sizeof(J));
    return 0;
  }
  if (rep == 3) {
    inv_depth_jac(p_FinA, J); /* :128-152 */
  } else if (rep == 4) {      /* :155-173 */
    const double rho = 1 / p_FinA[2], al = p_FinA[0] / p_FinA[2], be = p_FinA[1] / p_FinA[2];
    const double Jm[9] = {1.0 / rho, 0.0, -(1.0 / (rho * rho)) * al, 0.0, 1.0 / rho, -(1.0 / (rho * rho)) * be,
                          0.0, 0.0, -(1.0 / (rho * rho))};
    memcpy(J, Jm, sizeof(J));
  } else if (rep == 5) { /* :176-187 */
    const double rho = 1.0 / p_FinA[2];
    double dr[3];
    for (int k = 0; k < 3; ++k) dr[k] = -(1.0 / (rho * rho)) * (rho * p_FinA[k]);
    for (int i = 0; i < 3; ++i) dlam[i] = R_CtoG[3 * i] * dr[0] + R_CtoG[3 * i + 1] * dr[1] + R_CtoG[3 * i + 2] * dr[2];
    *nl_out = 1;
    return 0;
  } else {
    return -1;
  }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) dlam[3 * i + j] = R_CtoG[3 * i] * J[j] + R_CtoG[3 * i + 1] * J[3 + j] + R_CtoG[3 * i + 2] * J[6 + j];
  return 0;
}

/* get_feature_jacobian_full for a feature held in representation `rep` with anchor clone slot anchor_ci: the GLOBAL_3D
 * rows (H_f = w dz_dpfg) chained with the representation Jacobians, :411-421.  Column-major outputs like
 * ovo_feature_jacobian_full; H_f gets nl columns. */
int ovo_feature_jacobian_full_rep(const ovo_opts *o, const ovo_state *st, const ovo_feats *fb, int f, int rep, int anchor_ci,
                                  double *H_f, double *H_x, double *res, int *rows_out, int *cols_out, int *hf_cols_out,
                                  int *order_id, int *order_size, int *n_order_out) {
  int rows, cols, hfc, no;
  double *Hg = (double *)malloc(sizeof(double) * (size_t)(2 * fb->max_meas + 1) * 6);
  ovo_feats fbl = *fb;
  if (rep >= 2) fbl.p_FinG_fej = NULL; /* :299-302: the "best" global point also serves as first estimate */
  int rc = ovo_feature_jacobian_full(o, st, &fbl, f, o->sigma_constraint, 0, NULL, NULL, -1, Hg, H_x, res, &rows, &cols, &hfc,
                                     order_id, order_size, &no);
  if (rc) {
    free(Hg);
    return rc;
  }
  double dlam[9], H_anc[18], H_cal[18];
  int nl;
  rc = ovo_feature_jacobian_representation_fej(o, st, rep, fb->p_FinG + 3 * (size_t)f,
                                               fb->p_FinG_fej ? fb->p_FinG_fej + 3 * (size_t)f : NULL, anchor_ci, dlam, &nl, H_anc, H_cal);
  if (rc) {
    free(Hg);
    return rc;
  }
  for (int j = 0; j < nl; ++j)
    for (int i = 0; i < rows; ++i) {
      double a = 0;
      for (int k = 0; k < 3; ++k) a += CM(Hg, rows, i, k) * dlam[nl * k + j];
      CM(H_f, rows, i, j) = a;
    }
  if (rep >= 2) {
    int ct = 0;
    for (int v = 0; v < no; ++v) {
      const double *J = NULL;
      if (order_id[v] == st->clone_id[anchor_ci]) J = H_anc;
      else if (o->do_calib_camera_pose && order_id[v] == st->calib_id) J = H_cal;
      if (J)
        for (int j = 0; j < 6; ++j)
          for (int i = 0; i < rows; ++i) {
            double a = 0;
            for (int k = 0; k < 3; ++k) a += CM(Hg, rows, i, k) * J[6 * k + j];
            CM(H_x, rows, i, ct + j) += a;
          }
      ct += order_size[v];
    }
  }
  *rows_out = rows;
  *cols_out = cols;
  *hf_cols_out = nl;
  *n_order_out = no;
  free(Hg);
  return 0;
}

/* update/UpdaterHelper.cpp:515-546 ; update/UpdaterPlane.cpp:483-517 */
void ovo_nullspace_project(double *H_f, int rows, int hf_cols, double *H_x, int cols, double *H_cp, int cp_cols,
                           double *res) {
  for (int n = 0; n < hf_cols; ++n) {
    for (int m = rows - 1; m > n; --m) {
      double c, s;
      ovo_make_givens(CM(H_f, rows, m - 1, n), CM(H_f, rows, m, n), &c, &s);
      rot_rows(H_f, rows, m - 1, n, hf_cols, c, s);
      rot_rows(H_x, rows, m - 1, 0, cols, c, s);
      if (H_cp) rot_rows(H_cp, rows, m - 1, 0, cp_cols, c, s);
      rot_rows(res, rows, m - 1, 0, 1, c, s);
    }
  }
}

/* update/UpdaterHelper.cpp:548-579 ; update/UpdaterPlane.cpp:519-552 */
/* diagnostic tap (tools/plane_gate_study.py): state of the sweep in front of column n; not part of the restated algorithm */
typedef void (*ovo_compress_tap_fn)(int n, int rows, int cols, int ld, const double *H_x, const double *res);
static ovo_compress_tap_fn g_compress_tap = NULL;
void ovo_set_compress_tap(ovo_compress_tap_fn fn) { g_compress_tap = fn; }

int ovo_measurement_compress(double *H_x, int rows, int cols, int ld, double *H_cp, int cp_cols, int ld_cp,
                             double *res) {
  if (rows <= cols) return rows;
  for (int n = 0; n < cols; ++n) {
    if (g_compress_tap) g_compress_tap(n, rows, cols, ld, H_x, res);
    for (int m = rows - 1; m > n; --m) {
      double c, s;
      ovo_make_givens(CM(H_x, ld, m - 1, n), CM(H_x, ld, m, n), &c, &s);
      rot_rows(H_x, ld, m - 1, n, cols, c, s);
      if (H_cp) rot_rows(H_cp, ld_cp, m - 1, 0, cp_cols, c, s);
      rot_rows(res, rows, m - 1, 0, 1, c, s);
    }
  }
  return rows < cols ? rows : cols;
}

/* state/StateHelper.cpp:231-259 */
void ovo_marginal_cov(const double *P, int n, const int *order_id, const int *order_size, int n_order, double *out) {
  int cov_size = 0;
  for (int i = 0; i < n_order; ++i) cov_size += order_size[i];
  int i_index = 0;
  for (int i = 0; i < n_order; ++i) {
    int k_index = 0;
    for (int k = 0; k < n_order; ++k) {
      for (int cc = 0; cc < order_size[k]; ++cc)
        for (int rr = 0; rr < order_size[i]; ++rr)
          CM(out, cov_size, i_index + rr, k_index + cc) = CM(P, n, order_id[i] + rr, order_id[k] + cc);
      k_index += order_size[k];
    }
    i_index += order_size[i];
  }
}

/* state/StateHelper.cpp:121-202 (R = I) */
int ovo_ekf_update(double *P, int n, const int *order_id, const int *order_size, int n_order, const double *H,
                   int rows, int ld, const double *res, double *dx, int *neg_diag) {
  return ovo_ekf_update_rdiag(P, n, order_id, order_size, n_order, H, rows, ld, res, NULL, dx, neg_diag);
}

/* the same with R = diag(r_diag) instead of I (UpdaterZeroVelocity.cpp:265 hands a non-isotropic diagonal R) */
int ovo_ekf_update_rdiag(double *P, int n, const int *order_id, const int *order_size, int n_order, const double *H,
                         int rows, int ld, const double *res, const double *r_diag, double *dx, int *neg_diag) {
  int cols = 0;
  for (int i = 0; i < n_order; ++i) cols += order_size[i];
  int *gcol = (int *)malloc(sizeof(int) * (size_t)cols);
  {
    int t = 0;
    for (int i = 0; i < n_order; ++i)
      for (int k = 0; k < order_size[i]; ++k) gcol[t++] = order_id[i] + k;
  }
  /* M_a = P[:, gcol] * H^T   (n x rows) :142-151 */
  double *M = (double *)calloc((size_t)n * (size_t)rows, sizeof(double));
  for (int j = 0; j < rows; ++j)
    for (int k = 0; k < cols; ++k) {
      const double h = CM(H, ld, j, k);
      if (h == 0.0) continue;
      const double *pc = P + (size_t)gcol[k] * (size_t)n;
      double *mc = M + (size_t)j * (size_t)n;
      for (int r = 0; r < n; ++r) mc[r] += pc[r] * h;
    }
  /* S = H P_small H^T + I :156-161  ==  H * M[gcol,:] */
  double *S = (double *)calloc((size_t)rows * (size_t)rows, sizeof(double));
  for (int j = 0; j < rows; ++j)
    for (int k = 0; k < cols; ++k) {
      const double mkj = CM(M, n, gcol[k], j);
      for (int i = 0; i <= j; ++i) CM(S, rows, i, j) += CM(H, ld, i, k) * mkj;
    }
  for (int i = 0; i < rows; ++i) CM(S, rows, i, i) += r_diag ? r_diag[i] : 1.0;
  for (int j = 0; j < rows; ++j)
    for (int i = j + 1; i < rows; ++i) CM(S, rows, i, j) = CM(S, rows, j, i);
  /* Sinv via LLT :165-166 */
  int info = ovo_llt(S, rows, rows);
  if (info) {
    free(gcol);
    free(M);
    free(S);
    return -1;
  }
  double *Sinv = (double *)calloc((size_t)rows * (size_t)rows, sizeof(double));
  for (int j = 0; j < rows; ++j) {
    double *col = Sinv + (size_t)j * (size_t)rows;
    col[j] = 1.0;
    llt_solve_vec(S, rows, rows, col);
  }
  /* selfadjointView<Upper> of Sinv */
  for (int j = 0; j < rows; ++j)
    for (int i = j + 1; i < rows; ++i) CM(Sinv, rows, i, j) = CM(Sinv, rows, j, i);
  /* K = M Sinv :167 */
  double *K = (double *)calloc((size_t)n * (size_t)rows, sizeof(double));
  for (int j = 0; j < rows; ++j)
    for (int k = 0; k < rows; ++k) {
      const double sv = CM(Sinv, rows, k, j);
      const double *mc = M + (size_t)k * (size_t)n;
      double *kc = K + (size_t)j * (size_t)n;
      for (int r = 0; r < n; ++r) kc[r] += mc[r] * sv;
    }
  /* P.upper -= K M^T ; mirror :171-172 */
  for (int k = 0; k < rows; ++k) {
    const double *kc = K + (size_t)k * (size_t)n;
    const double *mc = M + (size_t)k * (size_t)n;
    for (int j = 0; j < n; ++j) {
      const double mj = mc[j];
      double *pc = P + (size_t)j * (size_t)n;
      for (int i = 0; i <= j; ++i) pc[i] -= kc[i] * mj;
    }
  }
  for (int j = 0; j < n; ++j)
    for (int i = j + 1; i < n; ++i) CM(P, n, i, j) = CM(P, n, j, i);
  *neg_diag = 0;
  for (int i = 0; i < n; ++i)
    if (CM(P, n, i, i) < 0.0) *neg_diag = 1;
  /* dx = K res :190 */
  for (int r = 0; r < n; ++r) dx[r] = 0.0;
  for (int k = 0; k < rows; ++k) {
    const double rv = res[k];
    const double *kc = K + (size_t)k * (size_t)n;
    for (int r = 0; r < n; ++r) dx[r] += kc[r] * rv;
  }
  free(gcol);
  free(M);
  free(S);
  free(Sinv);
  free(K);
  return 0;
}

/* state/StateHelper.cpp:41-119 */
int ovo_ekf_propagation(double *P, int n, int new_start, int phi_size, const int *old_id, const int *old_size,
                        int n_old, const double *Phi, const double *Q, int *neg_diag) {
  /* Phi is [phi_size x sum(old sizes)] column-major */
  double *CPT = (double *)calloc((size_t)n * (size_t)phi_size, sizeof(double));
  int loc = 0;
  for (int i = 0; i < n_old; ++i) {
    for (int k = 0; k < old_size[i]; ++k) {
      const double *pc = P + (size_t)(old_id[i] + k) * (size_t)n;
      for (int j = 0; j < phi_size; ++j) {
        const double ph = CM(Phi, phi_size, j, loc + k);
        double *cc = CPT + (size_t)j * (size_t)n;
        for (int r = 0; r < n; ++r) cc[r] += pc[r] * ph;
      }
    }
    loc += old_size[i];
  }
  double *PCP = (double *)calloc((size_t)phi_size * (size_t)phi_size, sizeof(double));
  for (int j = 0; j < phi_size; ++j)
    for (int i = 0; i < phi_size; ++i) CM(PCP, phi_size, i, j) = (i <= j) ? CM(Q, phi_size, i, j) : CM(Q, phi_size, j, i);
  loc = 0;
  for (int i = 0; i < n_old; ++i) {
    for (int k = 0; k < old_size[i]; ++k)
      for (int j = 0; j < phi_size; ++j) {
        const double cv = CM(CPT, n, old_id[i] + k, j);
        for (int r = 0; r < phi_size; ++r) CM(PCP, phi_size, r, j) += CM(Phi, phi_size, r, loc + k) * cv;
      }
    loc += old_size[i];
  }
  for (int j = 0; j < phi_size; ++j)
    for (int r = 0; r < n; ++r) {
      CM(P, n, new_start + j, r) = CM(CPT, n, r, j);
      CM(P, n, r, new_start + j) = CM(CPT, n, r, j);
    }
  for (int j = 0; j < phi_size; ++j)
    for (int i = 0; i < phi_size; ++i) CM(P, n, new_start + i, new_start + j) = CM(PCP, phi_size, i, j);
  *neg_diag = 0;
  for (int i = 0; i < n; ++i)
    if (CM(P, n, i, i) < 0.0) *neg_diag = 1;
  free(CPT);
  free(PCP);
  return 0;
}

/* ---------------------------------------------------------------------------------------------
 * update/UpdaterMSCKF.cpp:671-814
 * ------------------------------------------------------------------------------------------- */
int ovo_msckf_point_update(const ovo_opts *o, const ovo_state *st, const ovo_feats *fb, double *P, double *dx,
                           uint8_t *accepted, double *chi2_out, double *timings) {
  const double t0 = now_s();
  const int n = st->n_state;
  const int F = fb->n_feats;
  /* :671-691 sizes: 3 rows per measurement are reserved, full state width */
  size_t max_meas = 0;
  for (int f = 0; f < F; ++f) max_meas += 3 * (size_t)fb->n_meas[f];
  const int max_hx = n;
  double *res_big = (double *)calloc(max_meas ? max_meas : 1, sizeof(double));
  double *Hx_big = (double *)calloc((max_meas ? max_meas : 1) * (size_t)max_hx, sizeof(double));
  int *map_col = (int *)malloc(sizeof(int) * (size_t)n); /* Hx_mapping keyed by Type::id() */
  for (int i = 0; i < n; ++i) map_col[i] = -1;
  int *order_big_id = (int *)malloc(sizeof(int) * (size_t)(st->n_clones + 4));
  int *order_big_size = (int *)malloc(sizeof(int) * (size_t)(st->n_clones + 4));
  int n_order_big = 0;
  size_t ct_jacob = 0, ct_meas = 0;

  const int mm = fb->max_meas;
  const int maxrows = 3 * mm + 1, maxcols = 6 * mm + 14 + 3;
  double *H_f = (double *)malloc(sizeof(double) * (size_t)maxrows * 6);
  double *H_x = (double *)malloc(sizeof(double) * (size_t)maxrows * (size_t)maxcols);
  double *res = (double *)malloc(sizeof(double) * (size_t)maxrows);
  double *Pm = (double *)malloc(sizeof(double) * (size_t)maxcols * (size_t)maxcols);
  double *HP = (double *)malloc(sizeof(double) * (size_t)maxrows * (size_t)maxcols);
  double *S = (double *)malloc(sizeof(double) * (size_t)maxrows * (size_t)maxrows);
  double *tmp = (double *)malloc(sizeof(double) * (size_t)maxrows);
  int *oid = (int *)malloc(sizeof(int) * (size_t)(mm + 4));
  int *osz = (int *)malloc(sizeof(int) * (size_t)(mm + 4));

  for (int f = 0; f < F; ++f) {
    accepted[f] = 0;
    chi2_out[f] = 0.0;
    if (fb->n_meas[f] < 2) continue; /* UpdaterMSCKF.cpp:94-96 */
    int rows, cols, hfc, no;
    ovo_feature_jacobian_full(o, st, fb, f, o->sigma_constraint, 0, NULL, NULL, -1, H_f, H_x, res, &rows, &cols, &hfc,
                              oid, osz, &no);
    ovo_nullspace_project(H_f, rows, hfc, H_x, cols, NULL, 0, res); /* :736 */
    const int q = rows - hfc;
    /* :739-742 */
    ovo_marginal_cov(P, n, oid, osz, no, Pm);
    /* HP = Hp * Pm  (q x cols), Hp = rows [hfc, rows) of H_x */
    for (int j = 0; j < cols; ++j)
      for (int i = 0; i < q; ++i) CM(HP, q, i, j) = 0.0;
    for (int k = 0; k < cols; ++k)
      for (int j = 0; j < cols; ++j) {
        const double pv = CM(Pm, cols, k, j);
        for (int i = 0; i < q; ++i) CM(HP, q, i, j) += CM(H_x, rows, hfc + i, k) * pv;
      }
    for (int j = 0; j < q; ++j)
      for (int i = 0; i < q; ++i) CM(S, q, i, j) = (i == j) ? 1.0 : 0.0;
    for (int k = 0; k < cols; ++k)
      for (int j = 0; j < q; ++j) {
        const double hv = CM(H_x, rows, hfc + j, k);
        for (int i = 0; i < q; ++i) CM(S, q, i, j) += CM(HP, q, i, k) * hv;
      }
    if (ovo_llt(S, q, q)) continue;
    for (int i = 0; i < q; ++i) tmp[i] = res[hfc + i];
    llt_solve_vec(S, q, q, tmp);
    double chi2 = 0.0;
    for (int i = 0; i < q; ++i) chi2 += res[hfc + i] * tmp[i];
    chi2_out[f] = chi2;
    /* :745-764 */
    const double chi2_check = ovo_chi2_quantile_095(q);
    if (chi2 > o->chi2_multiplier * chi2_check) continue;
    accepted[f] = 1;
    /* :767-785 */
    int ct_hx = 0;
    for (int v = 0; v < no; ++v) {
      if (map_col[oid[v]] < 0) {
        map_col[oid[v]] = (int)ct_jacob;
        order_big_id[n_order_big] = oid[v];
        order_big_size[n_order_big++] = osz[v];
        ct_jacob += (size_t)osz[v];
      }
      for (int cc = 0; cc < osz[v]; ++cc)
        for (int i = 0; i < q; ++i)
          CM(Hx_big, max_meas, ct_meas + i, map_col[oid[v]] + cc) = CM(H_x, rows, hfc + i, ct_hx + cc);
      ct_hx += osz[v];
    }
    for (int i = 0; i < q; ++i) res_big[ct_meas + i] = res[hfc + i];
    ct_meas += (size_t)q;
  }
  const double t1 = now_s();
  int rows_c = 0;
  for (int i = 0; i < n; ++i) dx[i] = 0.0;
  double t2 = t1, t3 = t1;
  if (ct_meas >= 1) {
    /* :801-802 conservativeResize */
    double *Hc = (double *)malloc(sizeof(double) * ct_meas * (ct_jacob ? ct_jacob : 1));
    for (size_t j = 0; j < ct_jacob; ++j)
      memcpy(Hc + j * ct_meas, Hx_big + j * max_meas, sizeof(double) * ct_meas);
    /* :805 */
    rows_c = ovo_measurement_compress(Hc, (int)ct_meas, (int)ct_jacob, (int)ct_meas, NULL, 0, 0, res_big);
    t2 = now_s();
    /* :813-814 */
    int neg = 0;
    ovo_ekf_update(P, n, order_big_id, order_big_size, n_order_big, Hc, rows_c, (int)ct_meas, res_big, dx, &neg);
    t3 = now_s();
    free(Hc);
    if (neg) rows_c = -2;
  }
  if (timings) {
    timings[0] = t1 - t0;
    timings[1] = t2 - t1;
    timings[2] = t3 - t2;
    timings[3] = t3 - t0;
  }
  free(res_big);
  free(Hx_big);
  free(map_col);
  free(order_big_id);
  free(order_big_size);
  free(H_f);
  free(H_x);
  free(res);
  free(Pm);
  free(HP);
  free(S);
  free(tmp);
  free(oid);
  free(osz);
  return rows_c;
}

/* ---------------------------------------------------------------------------------------------
 * ext Type::update: JPL left-multiplicative quaternions, additive vectors
 * ------------------------------------------------------------------------------------------- */
void ovo_apply_dx(const ovo_state *st, int n_planes, const int *plane_state_id, const double *dx, ovo_state_values *val) {
  for (int i = 0; i < st->n_clones; ++i) {
    const int id = st->clone_id[i];
    ovo_quat_update(val->clone_q + 4 * i, dx + id);
    for (int k = 0; k < 3; ++k) val->clone_p[3 * i + k] += dx[id + 3 + k];
  }
  if (st->calib_id >= 0) {
    ovo_quat_update(val->calib_q, dx + st->calib_id);
    for (int k = 0; k < 3; ++k) val->calib_p[k] += dx[st->calib_id + 3 + k];
  }
  if (st->intr_id >= 0)
    for (int k = 0; k < 8; ++k) val->intrinsics[k] += dx[st->intr_id + k];
  for (int p = 0; p < n_planes; ++p)
    if (plane_state_id[p] >= 0)
      for (int k = 0; k < 3; ++k) val->cp[3 * p + k] += dx[plane_state_id[p] + k];
}

/* ---------------------------------------------------------------------------------------------
 * update/UpdaterMSCKF.cpp:411-649
 * ------------------------------------------------------------------------------------------- */
int ovo_msckf_plane_update(const ovo_opts *o, const ovo_state *st_in, const ovo_feats *fb, const int *plane_of_feat,
                           int n_planes, const double *cp_in, const double *cp_fej, const int *plane_state_id, double *P,
                           ovo_state_values *val, uint8_t *used, uint8_t *plane_ok, double *plane_chi2, int *plane_rows) {
  return ovo_msckf_plane_update_slam(o, st_in, fb, plane_of_feat, n_planes, cp_in, cp_fej, plane_state_id, P, val, used, plane_ok,
                                     plane_chi2, plane_rows, 0, NULL, NULL, NULL, NULL);
}

/* The same with the SLAM landmarks that lie on planes which are NOT in the state (update/UpdaterMSCKF.cpp:232-252): such a
 * feature has no measurements here, it contributes ONE point-on-plane row (update/UpdaterHelper.cpp:503-505) whose feature
 * Jacobian goes to the landmark's state columns instead of being projected away (:545-552).  slam_p is updated in place after
 * every accepted plane (the landmarks are state variables). */
/* Diagnostic tap (tools/plane_gate_study.py): called with the stacked system of a plane before the compression (stage 0:
 * rows x cols Hx, rows x 3 Hcp, res, all of leading dimension ld) and with the system handed to the chi2 test (stage 1: Hcp is
 * the marginal covariance cols x cols instead).  Not part of the restated algorithm. */
typedef void (*ovo_plane_tap_fn)(int plane, int stage, int rows, int cols, int ld, const double *Hx, const double *Hcp,
                                 const double *res);
static ovo_plane_tap_fn g_plane_tap = NULL;
void ovo_set_plane_tap(ovo_plane_tap_fn fn) { g_plane_tap = fn; }
/* Diagnostic: accept / reject sequence imposed from outside (one byte per plane, NULL = the gate decides), so that two builds of
 * this file can be compared plane by plane on the same sequence of states (tools/plane_gate_agreement.py). */
static const uint8_t *g_plane_force = NULL;
void ovo_set_plane_force(const uint8_t *force) { g_plane_force = force; }

int ovo_msckf_plane_update_slam(const ovo_opts *o, const ovo_state *st_in, const ovo_feats *fb, const int *plane_of_feat,
                                int n_planes, const double *cp_in, const double *cp_fej, const int *plane_state_id, double *P,
                                ovo_state_values *val, uint8_t *used, uint8_t *plane_ok, double *plane_chi2, int *plane_rows,
                                int n_slam, const int *slam_plane, const int *slam_id, double *slam_p, const double *slam_p_fej) {
  const int n = st_in->n_state;
  const int F = fb->n_feats;
  const int mm = fb->max_meas;
  for (int f = 0; f < F; ++f) used[f] = 0;
  (void)cp_in; /* current estimates live in val->cp */
  const int maxrows = 3 * mm + 1, maxcols = 6 * mm + 14 + 3;
  double *H_f_tmp = (double *)malloc(sizeof(double) * (size_t)maxrows * 6);
  double *H_f = (double *)malloc(sizeof(double) * (size_t)maxrows * 3);
  double *H_cp = (double *)malloc(sizeof(double) * (size_t)maxrows * 3);
  double *H_x_tmp = (double *)malloc(sizeof(double) * (size_t)maxrows * (size_t)maxcols);
  double *H_x = (double *)malloc(sizeof(double) * (size_t)maxrows * (size_t)maxcols);
  double *res = (double *)malloc(sizeof(double) * (size_t)maxrows);
  int *oid = (int *)malloc(sizeof(int) * (size_t)(mm + 4));
  int *osz = (int *)malloc(sizeof(int) * (size_t)(mm + 4));
  int *oid2 = (int *)malloc(sizeof(int) * (size_t)(mm + 4));
  int *osz2 = (int *)malloc(sizeof(int) * (size_t)(mm + 4));
  int *map_col = (int *)malloc(sizeof(int) * (size_t)n);
  int *order_big_id = (int *)malloc(sizeof(int) * (size_t)(st_in->n_clones + 8 + n_slam));
  int *order_big_size = (int *)malloc(sizeof(int) * (size_t)(st_in->n_clones + 8 + n_slam));
  double *dx = (double *)malloc(sizeof(double) * (size_t)n);

  for (int pl = 0; pl < n_planes; ++pl) {
    const int planeid = pl + 1;
    plane_ok[pl] = 0;
    plane_chi2[pl] = 0.0;
    plane_rows[pl] = 0;
    const int is_slam_plane = plane_state_id[pl] >= 0;
    /* features of this plane (MSCKF features only; UpdaterMSCKF.cpp:206-229) */
    int nf = 0;
    size_t max_meas = 0;
    for (int f = 0; f < F; ++f)
      if (plane_of_feat[f] == planeid && fb->n_meas[f] >= 2) {
        ++nf;
        max_meas += 3 * (size_t)fb->n_meas[f];
      }
    int ns = 0; /* SLAM landmarks on this plane: only planes outside the state collect them (:240-241) */
    if (!is_slam_plane)
      for (int q = 0; q < n_slam; ++q)
        if (slam_plane[q] == planeid) ++ns;
    max_meas += (size_t)ns;
    /* :316-317,384-396: a plane that is not in the state needs more than 3 features, at least one of them an MSCKF one */
    if (nf == 0 || (!is_slam_plane && nf + ns < 4)) continue;
    /* state tables at the current estimate (values change after every plane update, FEJ does not) */
    ovo_state st = *st_in;
    st.clone_q = val->clone_q;
    st.clone_p = val->clone_p;
    memcpy(st.calib_q, val->calib_q, sizeof(st.calib_q));
    memcpy(st.calib_p, val->calib_p, sizeof(st.calib_p));
    memcpy(st.intrinsics, val->intrinsics, sizeof(st.intrinsics));
    /* :467-475 plane linearisation points */
    const double *cpv = val->cp + 3 * pl;
    const double *cpf = is_slam_plane ? (cp_fej + 3 * pl) : cpv;

    double *res_big = (double *)calloc(max_meas, sizeof(double));
    double *Hx_big = (double *)calloc(max_meas * (size_t)n, sizeof(double));
    double *Hcp_big = (double *)calloc(max_meas * 3, sizeof(double));
    for (int i = 0; i < n; ++i) map_col[i] = -1;
    int n_order_big = 0;
    size_t ct_jacob = 0, ct_meas = 0;

    for (int f = 0; f < F; ++f) {
      if (plane_of_feat[f] != planeid || fb->n_meas[f] < 2) continue;
      int rows, cols, hfc, no;
      ovo_feature_jacobian_full(o, &st, fb, f, o->sigma_constraint, planeid, cpv, cpf, plane_state_id[pl], H_f_tmp,
                                H_x_tmp, res, &rows, &cols, &hfc, oid, osz, &no);
      int no2 = 0, cols2 = 0;
      if (is_slam_plane) {
        /* :518-535 pull the plane columns out of H_x */
        int ct_hx = 0, ct_new = 0;
        for (int v = 0; v < no; ++v) {
          if (oid[v] == plane_state_id[pl]) {
            memcpy(H_cp, H_x_tmp + (size_t)ct_hx * rows, sizeof(double) * (size_t)rows * 3);
          } else {
            oid2[no2] = oid[v];
            osz2[no2++] = osz[v];
            memcpy(H_x + (size_t)ct_new * rows, H_x_tmp + (size_t)ct_hx * rows, sizeof(double) * (size_t)rows * osz[v]);
            ct_new += osz[v];
          }
          ct_hx += osz[v];
        }
        cols2 = ct_new;
        memcpy(H_f, H_f_tmp, sizeof(double) * (size_t)rows * 3);
      } else {
        /* :537-540 */
        memcpy(H_cp, H_f_tmp + (size_t)3 * rows, sizeof(double) * (size_t)rows * 3);
        memcpy(H_f, H_f_tmp, sizeof(double) * (size_t)rows * 3);
        memcpy(H_x, H_x_tmp, sizeof(double) * (size_t)rows * cols);
        cols2 = cols;
        no2 = no;
        memcpy(oid2, oid, sizeof(int) * no);
        memcpy(osz2, osz, sizeof(int) * no);
      }
      /* :559 */
      ovo_nullspace_project(H_f, rows, 3, H_x, cols2, H_cp, 3, res);
      const int q = rows - 3;
      /* :564-576 */
      int ct_hx = 0;
      for (int v = 0; v < no2; ++v) {
        if (map_col[oid2[v]] < 0) {
          map_col[oid2[v]] = (int)ct_jacob;
          order_big_id[n_order_big] = oid2[v];
          order_big_size[n_order_big++] = osz2[v];
          ct_jacob += (size_t)osz2[v];
        }
        for (int cc = 0; cc < osz2[v]; ++cc)
          for (int i = 0; i < q; ++i)
            CM(Hx_big, max_meas, ct_meas + i, map_col[oid2[v]] + cc) = CM(H_x, rows, 3 + i, ct_hx + cc);
        ct_hx += osz2[v];
      }
      for (int cc = 0; cc < 3; ++cc)
        for (int i = 0; i < q; ++i) CM(Hcp_big, max_meas, ct_meas + i, cc) = CM(H_cp, rows, 3 + i, cc);
      for (int i = 0; i < q; ++i) res_big[ct_meas + i] = res[3 + i];
      ct_meas += (size_t)q;
    }
    /* SLAM landmarks of this plane: one constraint row each, landmark columns kept (:545-552) */
    for (int q = 0; q < n_slam && !is_slam_plane; ++q) {
      if (slam_plane[q] != planeid) continue;
      const double white_c = 1.0 / o->sigma_constraint;
      const double *pv = slam_p + 3 * q;
      const double *pj = o->do_fej ? slam_p_fej + 3 * q : pv; /* UpdaterHelper.cpp:467-477 */
      const double *cj = o->do_fej ? cpf : cpv;
      double d = sqrt(cpv[0] * cpv[0] + cpv[1] * cpv[1] + cpv[2] * cpv[2]);
      double nv[3] = {cpv[0] / d, cpv[1] / d, cpv[2] / d};
      const double r = white_c * (0.0 - ((nv[0] * pv[0] + nv[1] * pv[1] + nv[2] * pv[2]) - d));
      d = sqrt(cj[0] * cj[0] + cj[1] * cj[1] + cj[2] * cj[2]);
      nv[0] = cj[0] / d;
      nv[1] = cj[1] / d;
      nv[2] = cj[2] / d;
      const double np = nv[0] * pj[0] + nv[1] * pj[1] + nv[2] * pj[2];
      if (map_col[slam_id[q]] < 0) {
        map_col[slam_id[q]] = (int)ct_jacob;
        order_big_id[n_order_big] = slam_id[q];
        order_big_size[n_order_big++] = 3;
        ct_jacob += 3;
      }
      for (int cc = 0; cc < 3; ++cc) {
        CM(Hx_big, max_meas, ct_meas, map_col[slam_id[q]] + cc) = white_c * nv[cc];
        CM(Hcp_big, max_meas, ct_meas, cc) = white_c * 1.0 / d * (pj[cc] - np * nv[cc] - d * nv[cc]);
      }
      res_big[ct_meas] = r;
      ct_meas += 1;
    }
    /* :583-585 conservativeResize, :588 compress */
    const size_t hcols = ct_jacob + (is_slam_plane ? 3 : 0);
    double *Hc = (double *)malloc(sizeof(double) * ct_meas * (hcols ? hcols : 1));
    for (size_t j = 0; j < ct_jacob; ++j) memcpy(Hc + j * ct_meas, Hx_big + j * max_meas, sizeof(double) * ct_meas);
    double *Hcpc = (double *)malloc(sizeof(double) * ct_meas * 3);
    for (size_t j = 0; j < 3; ++j) memcpy(Hcpc + j * ct_meas, Hcp_big + j * max_meas, sizeof(double) * ct_meas);
    if (g_plane_tap) g_plane_tap(pl, 0, (int)ct_meas, (int)ct_jacob, (int)ct_meas, Hc, Hcpc, res_big);
    int rows_c = ovo_measurement_compress(Hc, (int)ct_meas, (int)ct_jacob, (int)ct_meas, Hcpc, 3, (int)ct_meas, res_big);
    int rows_u = rows_c;
    int row0 = 0; /* first row of the system handed to the chi2 test / update */
    if (is_slam_plane) {
      /* :593-600 append the plane columns */
      for (size_t j = 0; j < 3; ++j) memcpy(Hc + (ct_jacob + j) * ct_meas, Hcpc + j * ct_meas, sizeof(double) * ct_meas);
      order_big_id[n_order_big] = plane_state_id[pl];
      order_big_size[n_order_big++] = 3;
    } else {
      /* :602-603 project the plane out of the (compressed) system: rows [0, rows_c) of leading dimension ct_meas */
      /* work on a compact copy so that the Givens sweeps only see the retained rows */
      double *Hx2 = (double *)malloc(sizeof(double) * (size_t)rows_c * (ct_jacob ? ct_jacob : 1));
      double *Hcp2 = (double *)malloc(sizeof(double) * (size_t)rows_c * 3);
      double *res2 = (double *)malloc(sizeof(double) * (size_t)rows_c);
      for (size_t j = 0; j < ct_jacob; ++j)
        for (int i = 0; i < rows_c; ++i) CM(Hx2, rows_c, i, j) = CM(Hc, ct_meas, i, j);
      for (size_t j = 0; j < 3; ++j)
        for (int i = 0; i < rows_c; ++i) CM(Hcp2, rows_c, i, j) = CM(Hcpc, ct_meas, i, j);
      for (int i = 0; i < rows_c; ++i) res2[i] = res_big[i];
      ovo_nullspace_project(Hcp2, rows_c, 3, Hx2, (int)ct_jacob, NULL, 0, res2);
      rows_u = rows_c - 3;
      row0 = 0;
      for (size_t j = 0; j < ct_jacob; ++j)
        for (int i = 0; i < rows_u; ++i) CM(Hc, ct_meas, i, j) = CM(Hx2, rows_c, 3 + i, j);
      for (int i = 0; i < rows_u; ++i) res_big[i] = res2[3 + i];
      free(Hx2);
      free(Hcp2);
      free(res2);
    }
    (void)row0;
    /* :607-610 plane-level chi2 */
    const int hc = (int)hcols;
    double *Pm = (double *)malloc(sizeof(double) * (size_t)hc * hc);
    ovo_marginal_cov(P, n, order_big_id, order_big_size, n_order_big, Pm);
    if (g_plane_tap) g_plane_tap(pl, 1, rows_u, hc, (int)ct_meas, Hc, Pm, res_big);
    double *HP = (double *)calloc((size_t)rows_u * hc, sizeof(double));
    for (int k = 0; k < hc; ++k)
      for (int j = 0; j < hc; ++j) {
        const double pv = CM(Pm, hc, k, j);
        for (int i = 0; i < rows_u; ++i) CM(HP, rows_u, i, j) += CM(Hc, ct_meas, i, k) * pv;
      }
    double *S = (double *)calloc((size_t)rows_u * rows_u, sizeof(double));
    for (int i = 0; i < rows_u; ++i) CM(S, rows_u, i, i) = 1.0;
    for (int k = 0; k < hc; ++k)
      for (int j = 0; j < rows_u; ++j) {
        const double hv = CM(Hc, ct_meas, j, k);
        for (int i = 0; i < rows_u; ++i) CM(S, rows_u, i, j) += CM(HP, rows_u, i, k) * hv;
      }
    double chi2 = 0.0;
    int fail = ovo_llt(S, rows_u, rows_u);
    if (!fail) {
      double *tmp = (double *)malloc(sizeof(double) * (size_t)rows_u);
      memcpy(tmp, res_big, sizeof(double) * (size_t)rows_u);
      llt_solve_vec(S, rows_u, rows_u, tmp);
      for (int i = 0; i < rows_u; ++i) chi2 += res_big[i] * tmp[i];
      free(tmp);
    }
    plane_chi2[pl] = chi2;
    plane_rows[pl] = rows_u;
    const double chi2_check = ovo_chi2_quantile_095(rows_u);
    int accept = !fail && !(chi2 > o->chi2_multiplier * chi2_check);
    if (g_plane_force) accept = !fail && g_plane_force[pl] != 0;
    if (accept) {
      /* :635-648 */
      plane_ok[pl] = 1;
      for (int f = 0; f < F; ++f)
        if (plane_of_feat[f] == planeid && fb->n_meas[f] >= 2) used[f] = 1;
      int neg = 0;
      ovo_ekf_update(P, n, order_big_id, order_big_size, n_order_big, Hc, rows_u, (int)ct_meas, res_big, dx, &neg);
      ovo_apply_dx(st_in, n_planes, plane_state_id, dx, val);
      for (int q = 0; q < n_slam; ++q) /* ext Vec::update of every landmark */
        for (int a = 0; a < 3; ++a) slam_p[3 * q + a] += dx[slam_id[q] + a];
    }
    free(Pm);
    free(HP);
    free(S);
    free(Hc);
    free(Hcpc);
    free(res_big);
    free(Hx_big);
    free(Hcp_big);
  }
  free(H_f_tmp);
  free(H_f);
  free(H_cp);
  free(H_x_tmp);
  free(H_x);
  free(res);
  free(oid);
  free(osz);
  free(oid2);
  free(osz2);
  free(map_col);
  free(order_big_id);
  free(order_big_size);
  free(dx);
  return 0;
}

/* ---------------------------------------------------------------------------------------------
 * state/StateHelper.cpp:398-586
 * ------------------------------------------------------------------------------------------- */
/* inverse of a small k x k matrix (col-major) by Gauss-Jordan with partial pivoting
 * (the reference uses colPivHouseholderQr().inverse(), :564) */
static int small_inverse(const double *A, int k, double *Ainv) {
  double M[6 * 12];
  for (int i = 0; i < k; ++i)
    for (int j = 0; j < k; ++j) {
      M[i * 2 * k + j] = A[(size_t)j * k + i];
      M[i * 2 * k + k + j] = (i == j) ? 1.0 : 0.0;
    }
  for (int c = 0; c < k; ++c) {
    int piv = c;
    for (int r = c + 1; r < k; ++r)
      if (fabs(M[r * 2 * k + c]) > fabs(M[piv * 2 * k + c])) piv = r;
    if (M[piv * 2 * k + c] == 0.0) return -1;
    if (piv != c)
      for (int j = 0; j < 2 * k; ++j) {
        double t = M[c * 2 * k + j];
        M[c * 2 * k + j] = M[piv * 2 * k + j];
        M[piv * 2 * k + j] = t;
      }
    const double d = M[c * 2 * k + c];
    for (int j = 0; j < 2 * k; ++j) M[c * 2 * k + j] /= d;
    for (int r = 0; r < k; ++r)
      if (r != c) {
        const double f = M[r * 2 * k + c];
        for (int j = 0; j < 2 * k; ++j) M[r * 2 * k + j] -= f * M[c * 2 * k + j];
      }
  }
  for (int i = 0; i < k; ++i)
    for (int j = 0; j < k; ++j) Ainv[(size_t)j * k + i] = M[i * 2 * k + k + j];
  return 0;
}

int ovo_initialize(double *P, int n_cap, int *n_io, const int *order_id, const int *order_size, int n_order, double *H_R,
                   double *H_L, int rows, int k, double r_iso, double *res, double chi2_mult, int do_update,
                   double *new_var_delta, double *dx, double *chi2_out, int *dof_out) {
  const int n = *n_io;
  int cols = 0;
  for (int i = 0; i < n_order; ++i) cols += order_size[i];
  if (k > 6 || rows < k || n + k > n_cap) return -1;
  /* :434-446 Givens on H_L, applied to res and H_R */
  for (int c = 0; c < k; ++c)
    for (int m = rows - 1; m > c; --m) {
      double cs, sn;
      ovo_make_givens(CM(H_L, rows, m - 1, c), CM(H_L, rows, m, c), &cs, &sn);
      rot_rows(H_L, rows, m - 1, c, k, cs, sn);
      rot_rows(res, rows, m - 1, 0, 1, cs, sn);
      rot_rows(H_R, rows, m - 1, 0, cols, cs, sn);
    }
  const int rup = rows - k;
  int *gcol = (int *)malloc(sizeof(int) * (size_t)cols);
  {
    int t = 0;
    for (int i = 0; i < n_order; ++i)
      for (int q = 0; q < order_size[i]; ++q) gcol[t++] = order_id[i] + q;
  }
  /* :464-475 chi2 on the update part with the prior covariance, dof = res.rows() */
  double chi2 = 0.0;
  if (rup > 0) {
    double *HP = (double *)calloc((size_t)rup * cols, sizeof(double));
    for (int a = 0; a < cols; ++a)
      for (int b = 0; b < cols; ++b) {
        const double pv = CM(P, n_cap, gcol[a], gcol[b]);
        for (int i = 0; i < rup; ++i) CM(HP, rup, i, b) += CM(H_R, rows, k + i, a) * pv;
      }
    double *S = (double *)calloc((size_t)rup * rup, sizeof(double));
    for (int i = 0; i < rup; ++i) CM(S, rup, i, i) = r_iso;
    for (int a = 0; a < cols; ++a)
      for (int j = 0; j < rup; ++j) {
        const double hv = CM(H_R, rows, k + j, a);
        for (int i = 0; i < rup; ++i) CM(S, rup, i, j) += CM(HP, rup, i, a) * hv;
      }
    if (ovo_llt(S, rup, rup)) {
      free(HP);
      free(S);
      free(gcol);
      return -2;
    }
    double *tmp = (double *)malloc(sizeof(double) * (size_t)rup);
    for (int i = 0; i < rup; ++i) tmp[i] = res[k + i];
    llt_solve_vec(S, rup, rup, tmp);
    for (int i = 0; i < rup; ++i) chi2 += res[k + i] * tmp[i];
    free(tmp);
    free(HP);
    free(S);
  }
  if (chi2_out) *chi2_out = chi2;
  if (dof_out) *dof_out = rows;
  if (chi2 > chi2_mult * ovo_chi2_quantile_095(rows)) {
    free(gcol);
    return 0;
  }
  /* initialize_invertible :520-581 with Hxinit = top k rows of H_R, H_finit = top k x k of H_L */
  double *M_a = (double *)calloc((size_t)n * k, sizeof(double));
  for (int j = 0; j < k; ++j)
    for (int a = 0; a < cols; ++a) {
      const double hv = CM(H_R, rows, j, a);
      for (int r = 0; r < n; ++r) CM(M_a, n, r, j) += CM(P, n_cap, r, gcol[a]) * hv;
    }
  double Mm[36], HL[36], HLinv[36];
  for (int i = 0; i < k; ++i)
    for (int j = 0; j < k; ++j) {
      double s = (i == j) ? r_iso : 0.0;
      for (int a = 0; a < cols; ++a) s += CM(H_R, rows, i, a) * CM(M_a, n, gcol[a], j);
      Mm[(size_t)j * k + i] = s;
      HL[(size_t)j * k + i] = CM(H_L, rows, i, j);
    }
  /* selfadjointView<Upper> of M */
  for (int j = 0; j < k; ++j)
    for (int i = j + 1; i < k; ++i) Mm[(size_t)j * k + i] = Mm[(size_t)i * k + j];
  if (small_inverse(HL, k, HLinv)) {
    free(M_a);
    free(gcol);
    return -3;
  }
  /* P_LL = H_Linv M H_Linv^T ; cross = -M_a H_Linv^T */
  double T1[36], PLL[36];
  for (int i = 0; i < k; ++i)
    for (int j = 0; j < k; ++j) {
      double s = 0.0;
      for (int a = 0; a < k; ++a) s += HLinv[(size_t)a * k + i] * Mm[(size_t)j * k + a];
      T1[(size_t)j * k + i] = s;
    }
  for (int i = 0; i < k; ++i)
    for (int j = 0; j < k; ++j) {
      double s = 0.0;
      for (int a = 0; a < k; ++a) s += T1[(size_t)a * k + i] * HLinv[(size_t)a * k + j];
      PLL[(size_t)j * k + i] = s;
    }
  for (int r = 0; r < n; ++r)
    for (int j = 0; j < k; ++j) {
      double s = 0.0;
      for (int a = 0; a < k; ++a) s += CM(M_a, n, r, a) * HLinv[(size_t)a * k + j];
      CM(P, n_cap, r, n + j) = -s;
      CM(P, n_cap, n + j, r) = -s;
    }
  for (int i = 0; i < k; ++i)
    for (int j = 0; j < k; ++j) CM(P, n_cap, n + i, n + j) = PLL[(size_t)j * k + i];
  for (int i = 0; i < k; ++i) {
    double s = 0.0;
    for (int a = 0; a < k; ++a) s += HLinv[(size_t)a * k + i] * res[a];
    new_var_delta[i] = s; /* :577 */
  }
  free(M_a);
  *n_io = n + k;
  const int n2 = n + k;
  for (int i = 0; i < n2; ++i) dx[i] = 0.0;
  /* :483-485 update with the remaining rows (whitened: R = r_iso I) */
  if (rup > 0 && do_update) {
    /* compact copy of the update rows, whitened */
    const double w = 1.0 / sqrt(r_iso);
    double *Hup = (double *)malloc(sizeof(double) * (size_t)rup * cols);
    double *rupv = (double *)malloc(sizeof(double) * (size_t)rup);
    for (int a = 0; a < cols; ++a)
      for (int i = 0; i < rup; ++i) CM(Hup, rup, i, a) = w * CM(H_R, rows, k + i, a);
    for (int i = 0; i < rup; ++i) rupv[i] = w * res[k + i];
    /* ovo_ekf_update works on a dense n2 x n2 matrix with leading dimension n2: repack */
    double *Pd = (double *)malloc(sizeof(double) * (size_t)n2 * n2);
    for (int j = 0; j < n2; ++j)
      for (int i = 0; i < n2; ++i) CM(Pd, n2, i, j) = CM(P, n_cap, i, j);
    int neg = 0;
    ovo_ekf_update(Pd, n2, order_id, order_size, n_order, Hup, rup, rup, rupv, dx, &neg);
    for (int j = 0; j < n2; ++j)
      for (int i = 0; i < n2; ++i) CM(P, n_cap, i, j) = CM(Pd, n2, i, j);
    for (int i = 0; i < k; ++i) new_var_delta[i] += dx[n + i];
    free(Pd);
    free(Hup);
    free(rupv);
  }
  free(gcol);
  return 1;
}

/* ---------------------------------------------------------------------------------------------
 * update/UpdaterPlane.cpp:296-481
 * ------------------------------------------------------------------------------------------- */
int ovo_plane_init(const ovo_opts *o, const ovo_state *st_in, const ovo_feats *fb, const int *plane_of_feat, int n_planes,
                   const double *cp_in, double const_init_multi, double const_init_chi2, double *P, int n_cap, int *n_io,
                   ovo_state_values *val, uint8_t *used, uint8_t *plane_ok, double *plane_chi2, int *plane_dof,
                   int *new_id, double *cp_out) {
  const int F = fb->n_feats, mm = fb->max_meas;
  for (int f = 0; f < F; ++f) used[f] = 0;
  const int maxrows = 3 * mm + 1, maxcols = 6 * mm + 14 + 3;
  double *H_f_tmp = (double *)malloc(sizeof(double) * (size_t)maxrows * 6);
  double *H_f = (double *)malloc(sizeof(double) * (size_t)maxrows * 3);
  double *H_cp = (double *)malloc(sizeof(double) * (size_t)maxrows * 3);
  double *H_x = (double *)malloc(sizeof(double) * (size_t)maxrows * (size_t)maxcols);
  double *res = (double *)malloc(sizeof(double) * (size_t)maxrows);
  int *oid = (int *)malloc(sizeof(int) * (size_t)(mm + 4));
  int *osz = (int *)malloc(sizeof(int) * (size_t)(mm + 4));
  int *map_col = (int *)malloc(sizeof(int) * (size_t)n_cap);
  int *order_big_id = (int *)malloc(sizeof(int) * (size_t)(st_in->n_clones + 8));
  int *order_big_size = (int *)malloc(sizeof(int) * (size_t)(st_in->n_clones + 8));
  double *dx = (double *)malloc(sizeof(double) * (size_t)n_cap);
  for (int pl = 0; pl < n_planes; ++pl) {
    const int planeid = pl + 1;
    plane_ok[pl] = 0;
    plane_chi2[pl] = 0.0;
    plane_dof[pl] = 0;
    new_id[pl] = -1;
    for (int q = 0; q < 3; ++q) cp_out[3 * pl + q] = cp_in[3 * pl + q];
    const int n = *n_io;
    int nf = 0;
    size_t max_meas = 0;
    for (int f = 0; f < F; ++f)
      if (plane_of_feat[f] == planeid && fb->n_meas[f] >= 2) {
        ++nf;
        max_meas += 3 * (size_t)fb->n_meas[f];
      }
    if (nf < 3) continue; /* :303 */
    ovo_state st = *st_in;
    st.n_state = n;
    st.clone_q = val->clone_q;
    st.clone_p = val->clone_p;
    memcpy(st.calib_q, val->calib_q, sizeof(st.calib_q));
    memcpy(st.calib_p, val->calib_p, sizeof(st.calib_p));
    memcpy(st.intrinsics, val->intrinsics, sizeof(st.intrinsics));
    const double *cpv = cp_in + 3 * pl; /* :344-347 plane linearisation point = the estimate, fej = value */
    double *res_big = (double *)calloc(max_meas, sizeof(double));
    double *Hx_big = (double *)calloc(max_meas * (size_t)n, sizeof(double));
    double *Hcp_big = (double *)calloc(max_meas * 3, sizeof(double));
    for (int i = 0; i < n_cap; ++i) map_col[i] = -1;
    int n_order_big = 0;
    size_t ct_jacob = 0, ct_meas = 0;
    const double sigma_c = const_init_multi * o->sigma_constraint; /* :384 */
    for (int f = 0; f < F; ++f) {
      if (plane_of_feat[f] != planeid || fb->n_meas[f] < 2) continue;
      int rows, cols, hfc, no;
      ovo_feature_jacobian_full(o, &st, fb, f, sigma_c, planeid, cpv, cpv, -1, H_f_tmp, H_x, res, &rows, &cols, &hfc, oid, osz,
                                &no);
      memcpy(H_cp, H_f_tmp + (size_t)3 * rows, sizeof(double) * (size_t)rows * 3); /* :389 */
      memcpy(H_f, H_f_tmp, sizeof(double) * (size_t)rows * 3);
      ovo_nullspace_project(H_f, rows, 3, H_x, cols, H_cp, 3, res); /* :402 */
      const int q = rows - 3;
      int ct_hx = 0;
      for (int v = 0; v < no; ++v) {
        if (map_col[oid[v]] < 0) {
          map_col[oid[v]] = (int)ct_jacob;
          order_big_id[n_order_big] = oid[v];
          order_big_size[n_order_big++] = osz[v];
          ct_jacob += (size_t)osz[v];
        }
        for (int cc = 0; cc < osz[v]; ++cc)
          for (int i = 0; i < q; ++i)
            CM(Hx_big, max_meas, ct_meas + i, map_col[oid[v]] + cc) = CM(H_x, rows, 3 + i, ct_hx + cc);
        ct_hx += osz[v];
      }
      for (int cc = 0; cc < 3; ++cc)
        for (int i = 0; i < q; ++i) CM(Hcp_big, max_meas, ct_meas + i, cc) = CM(H_cp, rows, 3 + i, cc);
      for (int i = 0; i < q; ++i) res_big[ct_meas + i] = res[3 + i];
      ct_meas += (size_t)q;
    }
    /* :431-436 resize + compress */
    double *Hc = (double *)malloc(sizeof(double) * ct_meas * (ct_jacob ? ct_jacob : 1));
    for (size_t j = 0; j < ct_jacob; ++j) memcpy(Hc + j * ct_meas, Hx_big + j * max_meas, sizeof(double) * ct_meas);
    double *Hcpc = (double *)malloc(sizeof(double) * ct_meas * 3);
    for (size_t j = 0; j < 3; ++j) memcpy(Hcpc + j * ct_meas, Hcp_big + j * max_meas, sizeof(double) * ct_meas);
    const int rows_c = ovo_measurement_compress(Hc, (int)ct_meas, (int)ct_jacob, (int)ct_meas, Hcpc, 3, (int)ct_meas, res_big);
    /* compact (rows_c x .) copies for initialize */
    double *HR = (double *)malloc(sizeof(double) * (size_t)rows_c * (ct_jacob ? ct_jacob : 1));
    double *HL = (double *)malloc(sizeof(double) * (size_t)rows_c * 3);
    double *rr = (double *)malloc(sizeof(double) * (size_t)rows_c);
    for (size_t j = 0; j < ct_jacob; ++j)
      for (int i = 0; i < rows_c; ++i) CM(HR, rows_c, i, j) = CM(Hc, ct_meas, i, j);
    for (size_t j = 0; j < 3; ++j)
      for (int i = 0; i < rows_c; ++i) CM(HL, rows_c, i, j) = CM(Hcpc, ct_meas, i, j);
    for (int i = 0; i < rows_c; ++i) rr[i] = res_big[i];
    double delta[3], chi2 = 0.0;
    int dof = 0;
    int nn = n;
    const int ok = ovo_initialize(P, n_cap, &nn, order_big_id, order_big_size, n_order_big, HR, HL, rows_c, 3, 1.0, rr,
                                  const_init_chi2, 1, delta, dx, &chi2, &dof); /* :446 */
    plane_chi2[pl] = chi2;
    plane_dof[pl] = dof;
    if (ok == 1) {
      plane_ok[pl] = 1;
      new_id[pl] = n;
      *n_io = nn;
      for (int q = 0; q < 3; ++q) cp_out[3 * pl + q] = cp_in[3 * pl + q] + delta[q];
      /* planes initialised earlier in this call are state variables now and receive this update's correction */
      for (int g = 0; g < pl; ++g)
        if (new_id[g] >= 0)
          for (int q = 0; q < 3; ++q) cp_out[3 * g + q] += dx[new_id[g] + q];
      for (int f = 0; f < F; ++f)
        if (plane_of_feat[f] == planeid && fb->n_meas[f] >= 2) used[f] = 1;
      /* Type::update of the pre-existing variables with the EKF correction of the update rows */
      int no_planes = 0;
      ovo_apply_dx(st_in, no_planes, NULL, dx, val);
    }
    free(HR);
    free(HL);
    free(rr);
    free(Hc);
    free(Hcpc);
    free(res_big);
    free(Hx_big);
    free(Hcp_big);
  }
  free(H_f_tmp);
  free(H_f);
  free(H_cp);
  free(H_x);
  free(res);
  free(oid);
  free(osz);
  free(map_col);
  free(order_big_id);
  free(order_big_size);
  free(dx);
  return 0;
}

/* ---------------------------------------------------------------------------------------------
 * update/UpdaterSLAM.cpp:376-682
 * ------------------------------------------------------------------------------------------- */
static double chi2_of(const double *P, int n, const double *H, int rows, int cols, const int *oid, const int *osz, int no,
                      const double *res) {
  double *Pm = (double *)malloc(sizeof(double) * (size_t)cols * cols);
  ovo_marginal_cov(P, n, oid, osz, no, Pm);
  double *HP = (double *)calloc((size_t)rows * cols, sizeof(double));
  for (int k = 0; k < cols; ++k)
    for (int j = 0; j < cols; ++j) {
      const double pv = CM(Pm, cols, k, j);
      for (int i = 0; i < rows; ++i) CM(HP, rows, i, j) += CM(H, rows, i, k) * pv;
    }
  double *S = (double *)calloc((size_t)rows * rows, sizeof(double));
  for (int i = 0; i < rows; ++i) CM(S, rows, i, i) = 1.0;
  for (int k = 0; k < cols; ++k)
    for (int j = 0; j < rows; ++j) {
      const double hv = CM(H, rows, j, k);
      for (int i = 0; i < rows; ++i) CM(S, rows, i, j) += CM(HP, rows, i, k) * hv;
    }
  double chi2 = 1e300;
  if (!ovo_llt(S, rows, rows)) {
    double *tmp = (double *)malloc(sizeof(double) * (size_t)rows);
    memcpy(tmp, res, sizeof(double) * (size_t)rows);
    llt_solve_vec(S, rows, rows, tmp);
    chi2 = 0.0;
    for (int i = 0; i < rows; ++i) chi2 += res[i] * tmp[i];
    free(tmp);
  }
  free(Pm);
  free(HP);
  free(S);
  return chi2;
}

int ovo_slam_update(const ovo_opts *o, const ovo_state *st, const ovo_feats *fb, const int *lm_id, const int *plane_of_feat,
                    int n_planes, const double *cp, const double *cp_fej, const int *plane_state_id, double *P, double *dx,
                    uint8_t *accepted, double *chi2_out, uint8_t *fellback) {
  return ovo_slam_update_rep(o, st, fb, lm_id, plane_of_feat, n_planes, cp, cp_fej, plane_state_id, P, dx, accepted, chi2_out,
                             fellback, NULL, NULL);
}

/* the same with landmarks held in any ext LandmarkRepresentation: lm_rep[f] (0..5), lm_anchor[f] the anchor clone slot.
 * fb->p_FinG / p_FinG_fej stay the landmark's GLOBAL position (value / first estimate); dx on the landmark's columns is the
 * correction of its representation parameters.  Plane rows need GLOBAL_3D (update/UpdaterHelper.cpp:455-456). */
int ovo_slam_update_rep(const ovo_opts *o, const ovo_state *st, const ovo_feats *fb, const int *lm_id, const int *plane_of_feat,
                        int n_planes, const double *cp, const double *cp_fej, const int *plane_state_id, double *P, double *dx,
                        uint8_t *accepted, double *chi2_out, uint8_t *fellback, const int *lm_rep, const int *lm_anchor) {
  const int n = st->n_state, F = fb->n_feats, mm = fb->max_meas;
  size_t max_meas = 0;
  for (int f = 0; f < F; ++f) max_meas += 3 * (size_t)fb->n_meas[f];
  if (max_meas == 0) max_meas = 1;
  double *res_big = (double *)calloc(max_meas, sizeof(double));
  double *Hx_big = (double *)calloc(max_meas * (size_t)n, sizeof(double));
  int *map_col = (int *)malloc(sizeof(int) * (size_t)n);
  for (int i = 0; i < n; ++i) map_col[i] = -1;
  int *obig_id = (int *)malloc(sizeof(int) * (size_t)(st->n_clones + F + 8));
  int *obig_sz = (int *)malloc(sizeof(int) * (size_t)(st->n_clones + F + 8));
  int n_obig = 0;
  size_t ct_jacob = 0, ct_meas = 0;
  const int maxrows = 3 * mm + 1, maxcols = 6 * mm + 14 + 3 + 3;
  double *H_f = (double *)malloc(sizeof(double) * (size_t)maxrows * 6);
  double *H_x = (double *)malloc(sizeof(double) * (size_t)maxrows * (size_t)maxcols);
  double *H_xf = (double *)malloc(sizeof(double) * (size_t)maxrows * (size_t)maxcols);
  double *res = (double *)malloc(sizeof(double) * (size_t)maxrows);
  int *oid = (int *)malloc(sizeof(int) * (size_t)(mm + 6));
  int *osz = (int *)malloc(sizeof(int) * (size_t)(mm + 6));
  for (int i = 0; i < n; ++i) dx[i] = 0.0;
  for (int f = 0; f < F; ++f) {
    accepted[f] = 0;
    chi2_out[f] = 0.0;
    fellback[f] = 0;
    if (fb->n_meas[f] < 1) continue; /* :412-415 */
    if (lm_rep && lm_rep[f] == 5 && fb->n_meas[f] < 2) continue; /* :409-410, :416-418 required_meas = 2 */
    int planeid = 0, psid = -1;
    const double *cpv = NULL, *cpf = NULL;
    if (n_planes > 0 && plane_of_feat && plane_of_feat[f] > 0 && plane_state_id[plane_of_feat[f] - 1] >= 0) { /* :465-475 */
      planeid = plane_of_feat[f];
      psid = plane_state_id[planeid - 1];
      cpv = cp + 3 * (planeid - 1);
      cpf = cp_fej + 3 * (planeid - 1);
    }
    int rows = 0, cols = 0, hfc = 0, no = 0;
    double chi2 = 0.0;
    const int rep = lm_rep ? lm_rep[f] : 0;
    if (rep != 0 && planeid != 0) return -30;
    for (int attempt = 0; attempt < 2; ++attempt) {
      int nlm = 3;
      if (rep == 5) {
        /* :478-481 linearised as the MSCKF inverse depth; :499-515 the depth column joins the state side and the two bearing
         * columns are projected out of [H_x | depth | res] (UpdaterHelper::nullspace_project_inplace) */
        ovo_feature_jacobian_full_rep(o, st, fb, f, 4, lm_anchor[f], H_f, H_x, res, &rows, &cols, &hfc, oid, osz, &no);
        for (int i = 0; i < rows; ++i) CM(H_x, rows, i, cols) = CM(H_f, rows, i, 2);
        ovo_nullspace_project(H_f, rows, 2, H_x, cols + 1, NULL, 0, res);
        const int r2 = rows - 2;
        for (int j = 0; j <= cols; ++j)
          for (int i = 0; i < r2; ++i) CM(H_xf, r2, i, j) = CM(H_x, rows, i + 2, j);
        for (int j = 0; j < cols; ++j)
          for (int i = 0; i < r2; ++i) H_x[(size_t)j * r2 + i] = CM(H_xf, r2, i, j);
        for (int i = 0; i < r2; ++i) H_f[i] = CM(H_xf, r2, i, cols);
        for (int i = 0; i < r2; ++i) res[i] = res[i + 2];
        rows = r2;
        nlm = 1;
      } else if (rep != 0) {
        ovo_feature_jacobian_full_rep(o, st, fb, f, rep, lm_anchor[f], H_f, H_x, res, &rows, &cols, &hfc, oid, osz, &no);
        nlm = hfc;
      } else {
        ovo_feature_jacobian_full(o, st, fb, f, o->sigma_constraint, planeid, cpv, cpf, psid, H_f, H_x, res, &rows, &cols, &hfc,
                                  oid, osz, &no);
      }
      /* :517-522 append the landmark columns */
      memcpy(H_xf, H_x, sizeof(double) * (size_t)rows * cols);
      memcpy(H_xf + (size_t)rows * cols, H_f, sizeof(double) * (size_t)rows * nlm);
      oid[no] = lm_id[f];
      osz[no] = nlm;
      ++no;
      cols += nlm;
      chi2 = chi2_of(P, n, H_xf, rows, cols, oid, osz, no, res); /* :529-532 */
      const double thr = o->chi2_multiplier * ovo_chi2_quantile_095(rows);
      if (planeid != 0 && chi2 > thr) { /* :547-553 fallback without the plane */
        planeid = 0;
        psid = -1;
        fellback[f] = 1;
        continue;
      }
      break;
    }
    chi2_out[f] = chi2;
    if (chi2 > o->chi2_multiplier * ovo_chi2_quantile_095(rows)) continue; /* :596-619 */
    accepted[f] = 1;
    int ct_hx = 0;
    for (int v = 0; v < no; ++v) {
      if (map_col[oid[v]] < 0) {
        map_col[oid[v]] = (int)ct_jacob;
        obig_id[n_obig] = oid[v];
        obig_sz[n_obig++] = osz[v];
        ct_jacob += (size_t)osz[v];
      }
      for (int cc = 0; cc < osz[v]; ++cc)
        for (int i = 0; i < rows; ++i) CM(Hx_big, max_meas, ct_meas + i, map_col[oid[v]] + cc) = CM(H_xf, rows, i, ct_hx + cc);
      ct_hx += osz[v];
    }
    for (int i = 0; i < rows; ++i) res_big[ct_meas + i] = res[i];
    ct_meas += (size_t)rows;
  }
  int rc = 0;
  if (ct_meas >= 1) {
    double *Hc = (double *)malloc(sizeof(double) * ct_meas * (ct_jacob ? ct_jacob : 1));
    for (size_t j = 0; j < ct_jacob; ++j) memcpy(Hc + j * ct_meas, Hx_big + j * max_meas, sizeof(double) * ct_meas);
    int neg = 0;
    rc = ovo_ekf_update(P, n, obig_id, obig_sz, n_obig, Hc, (int)ct_meas, (int)ct_meas, res_big, dx, &neg); /* :673 */
    free(Hc);
  }
  free(res_big);
  free(Hx_big);
  free(map_col);
  free(obig_id);
  free(obig_sz);
  free(H_f);
  free(H_x);
  free(H_xf);
  free(res);
  free(oid);
  free(osz);
  return rc;
}

/* ---------------------------------------------------------------------------------------------
 * update/UpdaterSLAM.cpp:204-364
 * ------------------------------------------------------------------------------------------- */
int ovo_slam_delayed_init(const ovo_opts *o, const ovo_state *st_in, const ovo_feats *fb, double *P, int n_cap, int *n_io,
                          ovo_state_values *val, uint8_t *ok, double *chi2_out, int *new_id, double *p_out) {
  const int F = fb->n_feats, mm = fb->max_meas;
  const int maxrows = 3 * mm + 1, maxcols = 6 * mm + 14 + 3;
  double *H_f = (double *)malloc(sizeof(double) * (size_t)maxrows * 6);
  double *H_x = (double *)malloc(sizeof(double) * (size_t)maxrows * (size_t)maxcols);
  double *res = (double *)malloc(sizeof(double) * (size_t)maxrows);
  int *oid = (int *)malloc(sizeof(int) * (size_t)(mm + 6));
  int *osz = (int *)malloc(sizeof(int) * (size_t)(mm + 6));
  double *dx = (double *)malloc(sizeof(double) * (size_t)n_cap);
  for (int f = 0; f < F; ++f) {
    ok[f] = 0;
    chi2_out[f] = 0.0;
    new_id[f] = -1;
    for (int q = 0; q < 3; ++q) p_out[3 * f + q] = fb->p_FinG[3 * f + q];
    if (fb->n_meas[f] < 2) continue; /* :112-118 */
    ovo_state st = *st_in;
    st.n_state = *n_io;
    st.clone_q = val->clone_q;
    st.clone_p = val->clone_p;
    memcpy(st.calib_q, val->calib_q, sizeof(st.calib_q));
    memcpy(st.calib_p, val->calib_p, sizeof(st.calib_p));
    memcpy(st.intrinsics, val->intrinsics, sizeof(st.intrinsics));
    int rows, cols, hfc, no;
    ovo_feature_jacobian_full(o, &st, fb, f, o->sigma_constraint, 0, NULL, NULL, -1, H_f, H_x, res, &rows, &cols, &hfc, oid, osz, &no);
    double delta[3], chi2 = 0.0;
    int dof = 0, nn = *n_io;
    const int r = ovo_initialize(P, n_cap, &nn, oid, osz, no, H_x, H_f, rows, 3, 1.0, res, o->chi2_multiplier, 1, delta, dx, &chi2,
                                 &dof); /* :304 */
    chi2_out[f] = chi2;
    if (r == 1) {
      ok[f] = 1;
      new_id[f] = *n_io;
      *n_io = nn;
      for (int q = 0; q < 3; ++q) p_out[3 * f + q] = fb->p_FinG[3 * f + q] + delta[q];
      /* landmarks initialised earlier in this call are ordinary state variables now: the EKF update inside
       * StateHelper::initialize corrects them too (Type::update through StateHelper::EKFUpdate) */
      for (int g = 0; g < f; ++g)
        if (new_id[g] >= 0)
          for (int q = 0; q < 3; ++q) p_out[3 * g + q] += dx[new_id[g] + q];
      ovo_apply_dx(st_in, 0, NULL, dx, val);
    }
  }
  free(H_f);
  free(H_x);
  free(res);
  free(oid);
  free(osz);
  free(dx);
  return 0;
}

/* ---------------------------------------------------------------------------------------------
 * update/UpdaterZeroVelocity.cpp:68-318 with the constants the reference hard-codes (:113-116): the raw IMU readings of
 * [time0, time1] as measurements of "angular velocity = 0, specific force = R g" (:141-176), discrete noise sigma^2/dt
 * times zupt_noise_multiplier (:168-180), bias random walk dt * sigma (sic, :184-186) added to the 9 x 9 marginal for the chi2
 * (:191-194) and, when accepted, to P through EKFPropagation with Phi = I (:256-262), then EKFUpdate (:265).
 * imu_id = Type::id() of the IMU (error order th p v bg ba).  disparity_passed: outcome of :207-227 (the feature tracks are
 * not this function's input).  Returns 1 accepted / 0 rejected / <0 error; dx [n] = correction of the whole state.
 * ------------------------------------------------------------------------------------------- */
int ovo_zupt_update(const ovo_imu_state *x, const ovo_prop_opts *po, int imu_id, const double *imu, int n_imu, double time0,
                    double time1, double noise_multiplier, double chi2_multiplier, double max_velocity, int disparity_passed,
                    double *P, int n, double *dx, double *chi2_out, int *rows_out) {
  const int cap = n_imu + 4;
  double *sel = (double *)malloc(sizeof(double) * 7 * (size_t)cap);
  const int ns = ovo_select_imu_readings(imu, n_imu, time0, time1, sel, cap);
  *chi2_out = 0.0;
  *rows_out = 0;
  if (ns < 2) { /* :107-111 */
    free(sel);
    return 0;
  }
  const int ni = ns - 1, m = 6 * ni;
  double *H = (double *)calloc((size_t)m * 9, sizeof(double));
  double *res = (double *)calloc((size_t)m, sizeof(double));
  double *Rd = (double *)calloc((size_t)m, sizeof(double));
  double Rv[9], Rj[9], Rg[3], Rjg[3], S3[9];
  const double g[3] = {0.0, 0.0, po->gravity_mag};
  ovo_quat_2_rot(x->q, Rv);
  ovo_quat_2_rot(po->do_fej ? x->q_fej : x->q, Rj);
  mat3_vec(Rv, g, Rg);
  mat3_vec(Rj, g, Rjg);
  skew3(Rjg, S3);
  double dt_sum = 0.0;
  for (int i = 0; i < ni; ++i) {
    const double *r0 = sel + 7 * i, *r1 = sel + 7 * (i + 1);
    const double dt = r1[0] - r0[0];
    for (int k = 0; k < 3; ++k) {
      res[6 * i + k] = -(r0[1 + k] - x->bg[k]);
      res[6 * i + 3 + k] = -((r0[4 + k] - x->ba[k]) - Rg[k]);
      CM(H, m, 6 * i + k, 3 + k) = -1.0;
      for (int j = 0; j < 3; ++j) CM(H, m, 6 * i + 3 + k, j) = -S3[3 * k + j];
      CM(H, m, 6 * i + 3 + k, 6 + k) = -1.0;
      Rd[6 * i + k] = noise_multiplier * (po->sigma_w * po->sigma_w / dt);
      Rd[6 * i + 3 + k] = noise_multiplier * (po->sigma_a * po->sigma_a / dt);
    }
    dt_sum += dt;
  }
  double Qb[36];
  memset(Qb, 0, sizeof(Qb));
  for (int k = 0; k < 3; ++k) {
    Qb[7 * k] = dt_sum * po->sigma_wb;
    Qb[7 * (3 + k)] = dt_sum * po->sigma_ab;
  }
  const int oid[3] = {imu_id, imu_id + 9, imu_id + 12}, osz[3] = {3, 3, 3};
  double Pm[81];
  ovo_marginal_cov(P, n, oid, osz, 3, Pm);
  for (int a = 0; a < 6; ++a) CM(Pm, 9, 3 + a, 3 + a) += Qb[7 * a];
  /* S = H Pm H^T + R, chi2 = res^T S^-1 res */
  double *HP = (double *)calloc((size_t)m * 9, sizeof(double));
  double *S = (double *)calloc((size_t)m * (size_t)m, sizeof(double));
  for (int b = 0; b < 9; ++b)
    for (int a = 0; a < 9; ++a)
      for (int i = 0; i < m; ++i) CM(HP, m, i, b) += CM(H, m, i, a) * CM(Pm, 9, a, b);
  for (int j = 0; j < m; ++j)
    for (int a = 0; a < 9; ++a)
      for (int i = 0; i < m; ++i) CM(S, m, i, j) += CM(HP, m, i, a) * CM(H, m, j, a);
  for (int i = 0; i < m; ++i) CM(S, m, i, i) += Rd[i];
  int rc = 0;
  if (ovo_llt(S, m, m)) {
    rc = -1;
  } else {
    double *y = (double *)malloc(sizeof(double) * (size_t)m);
    memcpy(y, res, sizeof(double) * (size_t)m);
    llt_solve_vec(S, m, m, y);
    double c2 = 0.0;
    for (int i = 0; i < m; ++i) c2 += res[i] * y[i];
    free(y);
    *chi2_out = c2;
    *rows_out = m;
    const double vn = sqrt(x->v[0] * x->v[0] + x->v[1] * x->v[1] + x->v[2] * x->v[2]);
    if (!disparity_passed && (c2 > chi2_multiplier * ovo_chi2_quantile_095(m) || vn > max_velocity)) { /* :231-236 */
      rc = 0;
    } else {
      double Phi[36];
      memset(Phi, 0, sizeof(Phi));
      for (int k = 0; k < 6; ++k) Phi[7 * k] = 1.0;
      const int bid[2] = {imu_id + 9, imu_id + 12}, bsz[2] = {3, 3};
      int neg = 0;
      if (ovo_ekf_propagation(P, n, imu_id + 9, 6, bid, bsz, 2, Phi, Qb, &neg) || neg) rc = -2;
      else if (ovo_ekf_update_rdiag(P, n, oid, osz, 3, H, m, m, res, Rd, dx, &neg) || neg) rc = -3;
      else rc = 1;
    }
  }
  free(sel);
  free(H);
  free(res);
  free(Rd);
  free(HP);
  free(S);
  return rc;
}

/* ext Landmark::set_from_xyz (types/Landmark.cpp of ov_core, restated): representation parameters of a point given in the
 * frame the representation lives in (global for rep 0/1, anchor camera for rep 2..5). */
static void lm_params_from_xyz(int rep, const double p[3], double out[3]) {
  if (rep == 0 || rep == 2) {
    memcpy(out, p, 3 * sizeof(double));
  } else if (rep == 1 || rep == 3) {
    const double rho = 1.0 / sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
    out[0] = atan2(p[1], p[0]);
    out[1] = acos(rho * p[2]);
    out[2] = rho;
  } else if (rep == 4) {
    out[0] = p[0] / p[2];
    out[1] = p[1] / p[2];
    out[2] = 1.0 / p[2];
  } else { /* 5: only the inverse depth is a state; the bearing is kept aside (Landmark::_uv_norm_zero) */
    out[0] = 1.0 / p[2];
    out[1] = out[2] = 0.0;
  }
}

/* update/UpdaterSLAM.cpp:204-364 with StateOptions::feat_rep_slam = rep (:230-296).  A relative landmark is anchored in the
 * camera of its LAST measurement (ext FeatureInitializer::single_triangulation: anchor_clone_timestamp = timestamps.back()),
 * and triangulation hands over p_FinA: it is derived here from fb->p_FinG with the poses on entry, and stays fixed while the
 * poses are corrected by the earlier initialisations of the same call.  rep 5 (ANCHORED_INVERSE_DEPTH_SINGLE): Jacobians
 * of rep 4, depth column moved to the state side, bearing columns projected out, a 1-d variable is initialised (:262-283).
 * p_out [3*F]: representation parameters (rep 5: p_out[3f] only). */
int ovo_slam_delayed_init_rep(const ovo_opts *o, const ovo_state *st_in, const ovo_feats *fb, int rep, double *P, int n_cap,
                              int *n_io, ovo_state_values *val, uint8_t *ok, double *chi2_out, int *new_id, double *p_out) {
  const int F = fb->n_feats, mm = fb->max_meas;
  const int maxrows = 3 * mm + 1, maxcols = 6 * mm + 14 + 3 + 1;
  const int relative = rep >= 2, k_new = (rep == 5) ? 1 : 3, jrep = (rep == 5) ? 4 : rep;
  double *H_f = (double *)malloc(sizeof(double) * (size_t)maxrows * 6);
  double *H_x = (double *)malloc(sizeof(double) * (size_t)maxrows * (size_t)maxcols);
  double *H_x2 = (double *)malloc(sizeof(double) * (size_t)maxrows * (size_t)maxcols);
  double *res = (double *)malloc(sizeof(double) * (size_t)maxrows);
  int *oid = (int *)malloc(sizeof(int) * (size_t)(mm + 6));
  int *osz = (int *)malloc(sizeof(int) * (size_t)(mm + 6));
  double *dx = (double *)malloc(sizeof(double) * (size_t)n_cap);
  double *pA = (double *)malloc(sizeof(double) * 3 * (size_t)F);
  double *pG = (double *)malloc(sizeof(double) * 3 * (size_t)F);
  int *anc = (int *)malloc(sizeof(int) * (size_t)F);
  memcpy(pG, fb->p_FinG, sizeof(double) * 3 * (size_t)F);
  for (int f = 0; f < F; ++f) { /* what triangulation leaves on the Feature */
    anc[f] = fb->n_meas[f] > 0 ? fb->clone_idx[(size_t)f * mm + fb->n_meas[f] - 1] : 0;
    double R_ItoC[9], R_GtoI[9], d[3], t[3];
    ovo_quat_2_rot(val->calib_q, R_ItoC);
    ovo_quat_2_rot(val->clone_q + 4 * anc[f], R_GtoI);
    for (int q = 0; q < 3; ++q) d[q] = fb->p_FinG[3 * f + q] - val->clone_p[3 * anc[f] + q];
    mat3_vec(R_GtoI, d, t);
    mat3_vec(R_ItoC, t, pA + 3 * f);
    for (int q = 0; q < 3; ++q) pA[3 * f + q] += val->calib_p[q];
  }
  int rc_all = 0;
  for (int f = 0; f < F; ++f) {
    ok[f] = 0;
    chi2_out[f] = 0.0;
    new_id[f] = -1;
    lm_params_from_xyz(rep, relative ? pA + 3 * f : fb->p_FinG + 3 * f, p_out + 3 * f);
    if (fb->n_meas[f] < 2) continue; /* :112-118 */
    ovo_state st = *st_in;
    st.n_state = *n_io;
    st.clone_q = val->clone_q;
    st.clone_p = val->clone_p;
    memcpy(st.calib_q, val->calib_q, sizeof(st.calib_q));
    memcpy(st.calib_p, val->calib_p, sizeof(st.calib_p));
    memcpy(st.intrinsics, val->intrinsics, sizeof(st.intrinsics));
    if (relative) { /* UpdaterHelper.cpp:284-296: global position through the CURRENT anchor pose and calibration */
      double R_ItoC[9], R_GtoI[9], d[3], t[3];
      ovo_quat_2_rot(st.calib_q, R_ItoC);
      ovo_quat_2_rot(st.clone_q + 4 * anc[f], R_GtoI);
      for (int q = 0; q < 3; ++q) d[q] = pA[3 * f + q] - st.calib_p[q];
      for (int i = 0; i < 3; ++i) t[i] = R_ItoC[i] * d[0] + R_ItoC[3 + i] * d[1] + R_ItoC[6 + i] * d[2];
      for (int i = 0; i < 3; ++i)
        pG[3 * f + i] = R_GtoI[i] * t[0] + R_GtoI[3 + i] * t[1] + R_GtoI[6 + i] * t[2] + st.clone_p[3 * anc[f] + i];
    }
    ovo_feats fbl = *fb;
    fbl.p_FinG = pG;
    fbl.p_FinG_fej = NULL; /* :240-246 first estimate = value */
    int rows, cols, hfc, no;
    int rc = ovo_feature_jacobian_full_rep(o, &st, &fbl, f, jrep, anc[f], H_f, H_x, res, &rows, &cols, &hfc, oid, osz, &no);
    if (rc) {
      rc_all = rc;
      break;
    }
    double *HR = H_x, *HL = H_f;
    if (rep == 5) { /* :262-283 */
      for (int i = 0; i < rows; ++i) CM(H_x, rows, i, cols) = CM(H_f, rows, i, 2);
      ovo_nullspace_project(H_f, rows, 2, H_x, cols + 1, NULL, 0, res);
      const int r2 = rows - 2;
      for (int j = 0; j < cols; ++j)
        for (int i = 0; i < r2; ++i) CM(H_x2, r2, i, j) = CM(H_x, rows, i + 2, j);
      for (int i = 0; i < r2; ++i) H_f[i] = CM(H_x, rows, i + 2, cols);
      for (int i = 0; i < r2; ++i) res[i] = res[i + 2];
      rows = r2;
      HR = H_x2;
      HL = H_f;
    }
    double delta[3] = {0, 0, 0}, chi2 = 0.0;
    int dof = 0, nn = *n_io;
    const int r = ovo_initialize(P, n_cap, &nn, oid, osz, no, HR, HL, rows, k_new, 1.0, res, o->chi2_multiplier, 1, delta, dx, &chi2,
                                 &dof); /* :304 */
    chi2_out[f] = chi2;
    if (r == 1) {
      ok[f] = 1;
      new_id[f] = *n_io;
      *n_io = nn;
      for (int q = 0; q < k_new; ++q) p_out[3 * f + q] += delta[q];
      for (int g = 0; g < f; ++g)
        if (new_id[g] >= 0)
          for (int q = 0; q < k_new; ++q) p_out[3 * g + q] += dx[new_id[g] + q];
      ovo_apply_dx(st_in, 0, NULL, dx, val);
    }
  }
  free(H_f);
  free(H_x);
  free(H_x2);
  free(res);
  free(oid);
  free(osz);
  free(dx);
  free(pA);
  free(pG);
  free(anc);
  return rc_all;
}

/* ================================================================================================================
 * state/Propagator.cpp restatement (a11): IMU reading selection, mean integration, Phi / Qd accumulation.
 * ext quat_ops.h pieces: Omega, exp_so3, Jr_so3, quatnorm.
 * ============================================================================================================== */
static void pr_exp_so3(const double w[3], double R[9]) {
  double S[9], S2[9];
  skew3(w, S);
  mat3_mul(S, S, S2);
  const double th = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  double A, B;
  if (th < 1e-7) {
    A = 1.0;
    B = 0.5;
  } else {
    A = sin(th) / th;
    B = (1.0 - cos(th)) / (th * th);
  }
  for (int i = 0; i < 9; ++i) R[i] = A * S[i] + B * S2[i];
  R[0] += 1.0;
  R[4] += 1.0;
  R[8] += 1.0;
}

static void pr_jl_so3(const double w[3], double J[9]) {
  const double th = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  if (th < 1e-6) {
    for (int i = 0; i < 9; ++i) J[i] = 0.0;
    J[0] = J[4] = J[8] = 1.0;
    return;
  }
  const double a[3] = {w[0] / th, w[1] / th, w[2] / th};
  double S[9];
  skew3(a, S);
  const double s = sin(th) / th, c = (1.0 - cos(th)) / th;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) J[3 * i + j] = (i == j ? s : 0.0) + (1.0 - s) * a[i] * a[j] + c * S[3 * i + j];
}

static void pr_jr_so3(const double w[3], double J[9]) {
  const double m[3] = {-w[0], -w[1], -w[2]};
  pr_jl_so3(m, J);
}

/* Omega(w) q : [[-skew(w), w], [-w^T, 0]] applied to q (JPL) */
static void pr_omega_mul(const double w[3], const double q[4], double o[4]) {
  o[0] = w[2] * q[1] - w[1] * q[2] + w[0] * q[3];
  o[1] = -w[2] * q[0] + w[0] * q[2] + w[1] * q[3];
  o[2] = w[1] * q[0] - w[0] * q[1] + w[2] * q[3];
  o[3] = -w[0] * q[0] - w[1] * q[1] - w[2] * q[2];
}

static void pr_quatnorm(double q[4]) {
  if (q[3] < 0)
    for (int k = 0; k < 4; ++k) q[k] = -q[k];
  const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int k = 0; k < 4; ++k) q[k] /= n;
}

/* Propagator.cpp:571-591 (interpolate_data): linear in time */
static void pr_interp(const double *a, const double *b, double t, double *o) {
  const double lambda = (t - a[0]) / (b[0] - a[0]);
  o[0] = t;
  for (int k = 1; k < 7; ++k) o[k] = (1.0 - lambda) * a[k] + lambda * b[k];
}

/* Propagator.cpp:227-341.  imu [n x 7] rows = (t, wm, am).  Returns the number of selected readings (out [cap x 7]). */
int ovo_select_imu_readings(const double *imu, int n, double time0, double time1, double *out, int cap) {
  int m = 0;
  if (n == 0) return 0;
  for (int i = 0; i + 1 < n; ++i) {
    const double *a = imu + 7 * i, *b = imu + 7 * (i + 1);
    if (b[0] > time0 && a[0] < time0) { /* :246-252 start of the period: split */
      if (m < cap) pr_interp(a, b, time0, out + 7 * m);
      ++m;
      continue;
    }
    if (a[0] >= time0 && b[0] <= time1) { /* :257-261 middle */
      if (m < cap) memcpy(out + 7 * m, a, 7 * sizeof(double));
      ++m;
      continue;
    }
    if (b[0] > time1) { /* :268-296 end of the period */
      if (a[0] > time1 && i == 0) {
        break;
      } else if (a[0] > time1) {
        if (m < cap) pr_interp(imu + 7 * (i - 1), a, time1, out + 7 * m);
        ++m;
      } else {
        if (m < cap) memcpy(out + 7 * m, a, 7 * sizeof(double));
        ++m;
      }
      if (m <= cap && out[7 * (m - 1)] != time1) {
        if (m < cap) pr_interp(a, b, time1, out + 7 * m);
        ++m;
      }
      break;
    }
  }
  if (m == 0) return 0;
  if (m > cap) return -1;
  for (int i = 0; i + 1 < m; ++i) /* :316-325 zero dt */
    if (fabs(out[7 * (i + 1)] - out[7 * i]) < 1e-12) {
      memmove(out + 7 * i, out + 7 * (i + 1), sizeof(double) * 7 * (size_t)(m - i - 1));
      --m;
      --i;
    }
  return m;
}

/* Propagator.cpp:456-488 */
static void pr_mean_discrete(const ovo_imu_state *x, const ovo_prop_opts *po, double dt, const double *w1, const double *a1,
                             const double *w2, const double *a2, double *nq, double *nv, double *np) {
  double w[3], a[3];
  for (int k = 0; k < 3; ++k) {
    w[k] = po->imu_avg ? 0.5 * (w1[k] + w2[k]) : w1[k];
    a[k] = po->imu_avg ? 0.5 * (a1[k] + a2[k]) : a1[k];
  }
  const double wn = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  double R[9], Oq[4];
  ovo_quat_2_rot(x->q, R);
  pr_omega_mul(w, x->q, Oq);
  double ca, cb;
  if (wn > 1e-20) {
    ca = cos(0.5 * wn * dt);
    cb = 1.0 / wn * sin(0.5 * wn * dt);
  } else {
    ca = 1.0;
    cb = 0.5 * dt;
  }
  for (int k = 0; k < 4; ++k) nq[k] = ca * x->q[k] + cb * Oq[k];
  pr_quatnorm(nq);
  double Rta[3]; /* R^T a */
  for (int i = 0; i < 3; ++i) Rta[i] = R[i] * a[0] + R[3 + i] * a[1] + R[6 + i] * a[2];
  const double g[3] = {0.0, 0.0, po->gravity_mag};
  for (int i = 0; i < 3; ++i) {
    nv[i] = x->v[i] + Rta[i] * dt - g[i] * dt;
    np[i] = x->p[i] + x->v[i] * dt + 0.5 * Rta[i] * dt * dt - 0.5 * g[i] * dt * dt;
  }
}

/* Propagator.cpp:490-569 */
static void pr_mean_rk4(const ovo_imu_state *x, const ovo_prop_opts *po, double dt, const double *w1, const double *a1,
                        const double *w2, const double *a2, double *nq, double *nv, double *np) {
  double w[3], a[3], wal[3], aj[3];
  for (int k = 0; k < 3; ++k) {
    w[k] = w1[k];
    a[k] = a1[k];
    wal[k] = (w2[k] - w1[k]) / dt;
    aj[k] = (a2[k] - a1[k]) / dt;
  }
  const double g[3] = {0.0, 0.0, po->gravity_mag};
  double dq[4][4] = {{0, 0, 0, 1}}, kq[4][4], kp[4][3], kv[4][3], vs[3];
  for (int s = 0; s < 4; ++s) {
    if (s == 1 || s == 3)
      for (int k = 0; k < 3; ++k) {
        w[k] += 0.5 * wal[k] * dt;
        a[k] += 0.5 * aj[k] * dt;
      }
    if (s > 0) {
      const double f = (s == 3) ? 1.0 : 0.5;
      for (int k = 0; k < 4; ++k) dq[s][k] = dq[0][k] + f * kq[s - 1][k];
      pr_quatnorm(dq[s]);
      for (int k = 0; k < 3; ++k) vs[k] = x->v[k] + f * kv[s - 1][k];
    } else {
      for (int k = 0; k < 3; ++k) vs[k] = x->v[k];
    }
    double qd[4], qq[4], R[9];
    pr_omega_mul(w, dq[s], qd);
    quat_multiply(dq[s], x->q, qq);
    ovo_quat_2_rot(qq, R);
    for (int k = 0; k < 4; ++k) kq[s][k] = 0.5 * qd[k] * dt;
    for (int i = 0; i < 3; ++i) {
      kp[s][i] = vs[i] * dt;
      kv[s][i] = (R[i] * a[0] + R[3 + i] * a[1] + R[6 + i] * a[2] - g[i]) * dt;
    }
  }
  double dqf[4];
  for (int k = 0; k < 4; ++k)
    dqf[k] = dq[0][k] + (1.0 / 6.0) * kq[0][k] + (1.0 / 3.0) * kq[1][k] + (1.0 / 3.0) * kq[2][k] + (1.0 / 6.0) * kq[3][k];
  pr_quatnorm(dqf);
  quat_multiply(dqf, x->q, nq);
  for (int i = 0; i < 3; ++i) {
    np[i] = x->p[i] + (1.0 / 6.0) * kp[0][i] + (1.0 / 3.0) * kp[1][i] + (1.0 / 3.0) * kp[2][i] + (1.0 / 6.0) * kp[3][i];
    nv[i] = x->v[i] + (1.0 / 6.0) * kv[0][i] + (1.0 / 3.0) * kv[1][i] + (1.0 / 3.0) * kv[2][i] + (1.0 / 6.0) * kv[3][i];
  }
}

#define F15(i, j) F[(size_t)(j) * 15 + (i)]
#define G15(i, j) G[(size_t)(j) * 15 + (i)]
static void set33(double *M, int ld, int r0, int c0, const double *B /* row-major */, double s) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) M[(size_t)(c0 + j) * ld + r0 + i] = s * B[3 * i + j];
}

/* Propagator.cpp:343-454; x is advanced (value and FEJ := propagated value, :447-453). F, Qd col-major 15x15. */
void ovo_predict_and_compute(ovo_imu_state *x, const ovo_prop_opts *po, const double *minus, const double *plus, double *F,
                             double *Qd) {
  memset(F, 0, sizeof(double) * 225);
  memset(Qd, 0, sizeof(double) * 225);
  const double dt = plus[0] - minus[0];
  double w1[3], a1[3], w2[3], a2[3];
  for (int k = 0; k < 3; ++k) {
    w1[k] = minus[1 + k] - x->bg[k];
    a1[k] = minus[4 + k] - x->ba[k];
    w2[k] = plus[1 + k] - x->bg[k];
    a2[k] = plus[4 + k] - x->ba[k];
  }
  double nq[4], nv[3], np[3];
  if (po->use_rk4)
    pr_mean_rk4(x, po, dt, w1, a1, w2, a2, nq, nv, np);
  else
    pr_mean_discrete(x, po, dt, w1, a1, w2, a2, nq, nv, np);
  const int th = 0, p = 3, v = 6, bg = 9, ba = 12;
  const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  const double g[3] = {0.0, 0.0, po->gravity_mag};
  double G[15 * 12];
  memset(G, 0, sizeof(G));
  double mw[3] = {-w1[0] * dt, -w1[1] * dt, -w1[2] * dt}, Jr[9];
  pr_jr_so3(mw, Jr);
  if (po->do_fej) { /* :379-409 */
    double Rfej[9], Rn[9], RfT[9], dR[9], T[9], S[9], ST[9], a[3];
    ovo_quat_2_rot(x->q_fej, Rfej);
    ovo_quat_2_rot(nq, Rn);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) RfT[3 * i + j] = Rfej[3 * j + i];
    mat3_mul(Rn, RfT, dR);
    set33(F, 15, th, th, dR, 1.0);
    mat3_mul(dR, Jr, T);
    set33(F, 15, th, bg, T, -dt);
    set33(F, 15, bg, bg, I3, 1.0);
    for (int k = 0; k < 3; ++k) a[k] = nv[k] - x->v_fej[k] + g[k] * dt;
    skew3(a, S);
    mat3_mul(S, RfT, ST);
    set33(F, 15, v, th, ST, -1.0);
    set33(F, 15, v, v, I3, 1.0);
    set33(F, 15, v, ba, RfT, -dt);
    set33(F, 15, ba, ba, I3, 1.0);
    for (int k = 0; k < 3; ++k) a[k] = np[k] - x->p_fej[k] - x->v_fej[k] * dt + 0.5 * g[k] * dt * dt;
    skew3(a, S);
    mat3_mul(S, RfT, ST);
    set33(F, 15, p, th, ST, -1.0);
    set33(F, 15, p, v, I3, dt);
    set33(F, 15, p, ba, RfT, -0.5 * dt * dt);
    set33(F, 15, p, p, I3, 1.0);
    set33(G, 15, th, 0, T, -dt);
    set33(G, 15, v, 3, RfT, -dt);
    set33(G, 15, p, 3, RfT, -0.5 * dt * dt);
    set33(G, 15, bg, 6, I3, 1.0);
    set33(G, 15, ba, 9, I3, 1.0);
  } else { /* :411-432 */
    double R[9], RT[9], E[9], T[9], S[9], RS[9], a[3];
    ovo_quat_2_rot(x->q, R);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) RT[3 * i + j] = R[3 * j + i];
    pr_exp_so3(mw, E);
    set33(F, 15, th, th, E, 1.0);
    mat3_mul(E, Jr, T);
    set33(F, 15, th, bg, T, -dt);
    set33(F, 15, bg, bg, I3, 1.0);
    for (int k = 0; k < 3; ++k) a[k] = a1[k] * dt;
    skew3(a, S);
    mat3_mul(RT, S, RS);
    set33(F, 15, v, th, RS, -1.0);
    set33(F, 15, v, v, I3, 1.0);
    set33(F, 15, v, ba, RT, -dt);
    set33(F, 15, ba, ba, I3, 1.0);
    for (int k = 0; k < 3; ++k) a[k] = a1[k] * dt * dt;
    skew3(a, S);
    mat3_mul(RT, S, RS);
    set33(F, 15, p, th, RS, -0.5);
    set33(F, 15, p, v, I3, dt);
    set33(F, 15, p, ba, RT, -0.5 * dt * dt);
    set33(F, 15, p, p, I3, 1.0);
    set33(G, 15, th, 0, T, -dt);
    set33(G, 15, v, 3, RT, -dt);
    set33(G, 15, p, 3, RT, -0.5 * dt * dt);
    set33(G, 15, bg, 6, I3, 1.0);
    set33(G, 15, ba, 9, I3, 1.0);
  }
  /* :437-445 Qd = G Qc G^T, symmetrised */
  const double qc[4] = {po->sigma_w * po->sigma_w / dt, po->sigma_a * po->sigma_a / dt, po->sigma_wb * po->sigma_wb * dt,
                        po->sigma_ab * po->sigma_ab * dt};
  double T2[225];
  for (int i = 0; i < 15; ++i)
    for (int j = 0; j < 15; ++j) {
      double s = 0.0;
      for (int k = 0; k < 12; ++k) s += G15(i, k) * qc[k / 3] * G15(j, k);
      T2[(size_t)j * 15 + i] = s;
    }
  for (int i = 0; i < 15; ++i)
    for (int j = 0; j < 15; ++j) Qd[(size_t)j * 15 + i] = 0.5 * (T2[(size_t)j * 15 + i] + T2[(size_t)i * 15 + j]);
  /* :447-453 */
  memcpy(x->q, nq, sizeof(nq));
  memcpy(x->p, np, sizeof(np));
  memcpy(x->v, nv, sizeof(nv));
  memcpy(x->q_fej, nq, sizeof(nq));
  memcpy(x->p_fej, np, sizeof(np));
  memcpy(x->v_fej, nv, sizeof(nv));
  memcpy(x->bg_fej, x->bg, sizeof(x->bg));
  memcpy(x->ba_fej, x->ba, sizeof(x->ba));
}

/* Propagator.cpp:37-126 up to (not including) the covariance call: returns Phi_summed, Qd_summed (col-major 15x15),
 * the propagated IMU state and last_w.  The caller applies ovo_ekf_propagation with order {imu} and clones. */
int ovo_propagate_summed(ovo_imu_state *x, const ovo_prop_opts *po, const double *imu, int n_imu, double time0, double time1,
                         double *Phi, double *Qs, double *last_w, int *n_sel) {
  const int cap = n_imu + 4;
  double *sel = (double *)malloc(sizeof(double) * 7 * (size_t)cap);
  const int m = ovo_select_imu_readings(imu, n_imu, time0, time1, sel, cap);
  if (m < 0) {
    free(sel);
    return -1;
  }
  memset(Phi, 0, sizeof(double) * 225);
  memset(Qs, 0, sizeof(double) * 225);
  for (int i = 0; i < 15; ++i) Phi[(size_t)i * 15 + i] = 1.0;
  double F[225], Qdi[225], T[225], T2[225];
  for (int k = 0; k < 3; ++k) last_w[k] = 0.0;
  /* last_w uses the bias BEFORE... the bias is not propagated, so any time works (:110-114) */
  if (m > 1) {
    for (int i = 0; i + 1 < m; ++i) {
      ovo_predict_and_compute(x, po, sel + 7 * i, sel + 7 * (i + 1), F, Qdi);
      for (int r = 0; r < 15; ++r) /* Phi = F Phi */
        for (int c = 0; c < 15; ++c) {
          double s = 0.0;
          for (int k = 0; k < 15; ++k) s += F[(size_t)k * 15 + r] * Phi[(size_t)c * 15 + k];
          T[(size_t)c * 15 + r] = s;
        }
      memcpy(Phi, T, sizeof(T));
      for (int r = 0; r < 15; ++r) /* T = F Qs */
        for (int c = 0; c < 15; ++c) {
          double s = 0.0;
          for (int k = 0; k < 15; ++k) s += F[(size_t)k * 15 + r] * Qs[(size_t)c * 15 + k];
          T[(size_t)c * 15 + r] = s;
        }
      for (int r = 0; r < 15; ++r) /* T2 = T F^T + Qdi */
        for (int c = 0; c < 15; ++c) {
          double s = 0.0;
          for (int k = 0; k < 15; ++k) s += T[(size_t)k * 15 + r] * F[(size_t)k * 15 + c];
          T2[(size_t)c * 15 + r] = s + Qdi[(size_t)c * 15 + r];
        }
      for (int r = 0; r < 15; ++r)
        for (int c = 0; c < 15; ++c) Qs[(size_t)c * 15 + r] = 0.5 * (T2[(size_t)c * 15 + r] + T2[(size_t)r * 15 + c]);
    }
    for (int k = 0; k < 3; ++k) last_w[k] = sel[7 * (m - 2) + 1 + k] - x->bg[k];
  } else if (m == 1) {
    for (int k = 0; k < 3; ++k) last_w[k] = sel[1 + k] - x->bg[k];
  }
  *n_sel = m;
  free(sel);
  return 0;
}

/* ================================================================================================================
 * ext ov_core::FeatureInitializer::single_triangulation + single_gaussnewton, as called from
 * update/UpdaterMSCKF.cpp:120-166 (camera clone poses :123-135, mono).  The source is NOT in /root/reference (open_vins
 * ov_core/src/feat/FeatureInitializer.cpp @ 74a63cf): this is a restatement of the published algorithm from memory -
 * PARITY UNPINNED against the real implementation.  Kept faithful to what is remembered of its arithmetic: sequential sums
 * over the measurements, residuals of the refinement formed in single precision (Eigen::Matrix<float,2,1>).
 * ============================================================================================================== */
static void tri_sym3_eig(const double A[9], double ev[3]) { /* cyclic Jacobi on a symmetric 3x3, eigenvalues only */
  double a[9];
  memcpy(a, A, sizeof(a));
  for (int sweep = 0; sweep < 30; ++sweep) {
    const double off = fabs(a[1]) + fabs(a[2]) + fabs(a[5]);
    /* converged: the off-diagonal part is below the rounding of the diagonal (cyclic Jacobi converges quadratically - four or five
     * sweeps; the bound of 30 is never reached).  The eigenvalues only feed the condition-number test of the triangulation. */
    if (off <= 1e-17 * (fabs(a[0]) + fabs(a[4]) + fabs(a[8]))) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        const double apq = a[3 * p + q];
        if (apq == 0.0) continue;
        const double theta = (a[3 * q + q] - a[3 * p + p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; ++k) { /* columns p,q */
          const double akp = a[3 * k + p], akq = a[3 * k + q];
          a[3 * k + p] = c * akp - s * akq;
          a[3 * k + q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; ++k) { /* rows p,q */
          const double apk = a[3 * p + k], aqk = a[3 * q + k];
          a[3 * p + k] = c * apk - s * aqk;
          a[3 * q + k] = s * apk + c * aqk;
        }
      }
  }
  ev[0] = a[0];
  ev[1] = a[4];
  ev[2] = a[8];
}

/* x = A^-1 b for a 3x3 system, Gaussian elimination with full pivoting (the reference uses colPivHouseholderQr; both are
 * backward stable, the solutions agree to a few ulp) */
static int tri_solve3(const double A[9], const double b[3], double x[3]) {
  double m[12];
  int perm[3] = {0, 1, 2};
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) m[4 * i + j] = A[3 * i + j];
    m[4 * i + 3] = b[i];
  }
  for (int k = 0; k < 3; ++k) {
    int pi = k, pj = k;
    double best = -1.0;
    for (int i = k; i < 3; ++i)
      for (int j = k; j < 3; ++j)
        if (fabs(m[4 * i + j]) > best) {
          best = fabs(m[4 * i + j]);
          pi = i;
          pj = j;
        }
    if (!(best > 0.0)) return 1;
    if (pi != k)
      for (int j = 0; j < 4; ++j) {
        const double t = m[4 * k + j];
        m[4 * k + j] = m[4 * pi + j];
        m[4 * pi + j] = t;
      }
    if (pj != k) {
      for (int i = 0; i < 3; ++i) {
        const double t = m[4 * i + k];
        m[4 * i + k] = m[4 * i + pj];
        m[4 * i + pj] = t;
      }
      const int t = perm[k];
      perm[k] = perm[pj];
      perm[pj] = t;
    }
    for (int i = k + 1; i < 3; ++i) {
      const double f = m[4 * i + k] / m[4 * k + k];
      for (int j = k; j < 4; ++j) m[4 * i + j] -= f * m[4 * k + j];
    }
  }
  double y[3];
  for (int i = 2; i >= 0; --i) {
    double s = m[4 * i + 3];
    for (int j = i + 1; j < 3; ++j) s -= m[4 * i + j] * y[j];
    y[i] = s / m[4 * i + i];
  }
  for (int i = 0; i < 3; ++i) x[perm[i]] = y[i];
  return 0;
}

typedef struct {
  double R_AtoC[9], p_CinA[3], p_AinC[3];
} tri_rel;

static double tri_cost(const tri_rel *rel, const float *uvn, int m, double alpha, double beta, double rho) {
  double err = 0.0;
  for (int k = 0; k < m; ++k) {
    const double *R = rel[k].R_AtoC, *p = rel[k].p_AinC;
    const double hi1 = R[0] * alpha + R[1] * beta + R[2] + rho * p[0];
    const double hi2 = R[3] * alpha + R[4] * beta + R[5] + rho * p[1];
    const double hi3 = R[6] * alpha + R[7] * beta + R[8] + rho * p[2];
    const float z0 = (float)(hi1 / hi3), z1 = (float)(hi2 / hi3);
    const float r0 = uvn[2 * k] - z0, r1 = uvn[2 * k + 1] - z1;
    const float nrm = sqrtf(r0 * r0 + r1 * r1);
    err += (double)nrm * (double)nrm;
  }
  return err;
}

void ovo_triang_defaults(ovo_triang_opts *o) {
  o->refine_features = 1;
  o->max_runs = 5;
  o->init_lamda = 1e-3;
  o->max_lamda = 1e10;
  o->min_dx = 1e-6;
  o->min_dcost = 1e-6;
  o->lam_mult = 10.0;
  o->min_dist = 0.10;
  o->max_dist = 60.0;
  o->max_baseline = 40.0;
  o->max_cond_number = 10000.0;
  o->triangulate_1d = 0;
  o->reserved = 0;
}

int ovo_triangulate(const ovo_triang_opts *o, const ovo_state *st, const ovo_feats *fb, const float *uv_norm,
                    double *p_FinG_out, uint8_t *ok) {
  const int F = fb->n_feats, M = fb->max_meas, C = st->n_clones;
  /* camera poses of the clones, update/UpdaterMSCKF.cpp:123-135 */
  double *Rc = (double *)malloc(sizeof(double) * 9 * (size_t)C), *pc = (double *)malloc(sizeof(double) * 3 * (size_t)C);
  double R_ItoC[9];
  ovo_quat_2_rot(st->calib_q, R_ItoC);
  for (int i = 0; i < C; ++i) {
    double R_GtoI[9];
    ovo_quat_2_rot(st->clone_q + 4 * i, R_GtoI);
    mat3_mul(R_ItoC, R_GtoI, Rc + 9 * i);
    for (int a = 0; a < 3; ++a) /* p_CinG = p_IinG - R_GtoC^T p_IinC */
      pc[3 * i + a] = st->clone_p[3 * i + a] - (Rc[9 * i + a] * st->calib_p[0] + Rc[9 * i + 3 + a] * st->calib_p[1] +
                                                 Rc[9 * i + 6 + a] * st->calib_p[2]);
  }
  tri_rel *rel = (tri_rel *)malloc(sizeof(tri_rel) * (size_t)M);
  for (int f = 0; f < F; ++f) {
    ok[f] = 0;
    for (int a = 0; a < 3; ++a) p_FinG_out[3 * f + a] = 0.0;
    const int m = fb->n_meas[f];
    if (m < 2) continue;
    const int *ci = fb->clone_idx + (size_t)f * M;
    const float *uvn = uv_norm + (size_t)f * M * 2;
    /* anchor = last measurement of the (only) camera */
    const double *R_GtoA = Rc + 9 * ci[m - 1], *p_AinG = pc + 3 * ci[m - 1];
    double A[9] = {0}, b[3] = {0};
    for (int k = 0; k < m; ++k) {
      const double *R_GtoCi = Rc + 9 * ci[k], *p_CiinG = pc + 3 * ci[k];
      tri_rel *r = rel + k;
      for (int i = 0; i < 3; ++i) /* R_AtoCi = R_GtoCi R_GtoA^T */
        for (int j = 0; j < 3; ++j)
          r->R_AtoC[3 * i + j] = R_GtoCi[3 * i] * R_GtoA[3 * j] + R_GtoCi[3 * i + 1] * R_GtoA[3 * j + 1] + R_GtoCi[3 * i + 2] * R_GtoA[3 * j + 2];
      const double d[3] = {p_CiinG[0] - p_AinG[0], p_CiinG[1] - p_AinG[1], p_CiinG[2] - p_AinG[2]};
      mat3_vec(R_GtoA, d, r->p_CinA);
      double t[3];
      mat3_vec(r->R_AtoC, r->p_CinA, t);
      for (int a = 0; a < 3; ++a) r->p_AinC[a] = -t[a];
      /* bearing in the anchor frame */
      const double bc[3] = {(double)uvn[2 * k], (double)uvn[2 * k + 1], 1.0};
      double bi[3];
      for (int a = 0; a < 3; ++a) bi[a] = r->R_AtoC[a] * bc[0] + r->R_AtoC[3 + a] * bc[1] + r->R_AtoC[6 + a] * bc[2];
      const double nb = sqrt(bi[0] * bi[0] + bi[1] * bi[1] + bi[2] * bi[2]);
      for (int a = 0; a < 3; ++a) bi[a] /= nb;
      double S[9], Ai[9];
      skew3(bi, S);
      for (int i = 0; i < 3; ++i) /* Ai = S^T S */
        for (int j = 0; j < 3; ++j) Ai[3 * i + j] = S[i] * S[j] + S[3 + i] * S[3 + j] + S[6 + i] * S[6 + j];
      for (int i = 0; i < 9; ++i) A[i] += Ai[i];
      for (int i = 0; i < 3; ++i) b[i] += Ai[3 * i] * r->p_CinA[0] + Ai[3 * i + 1] * r->p_CinA[1] + Ai[3 * i + 2] * r->p_CinA[2];
    }
    double pA[3], ev[3];
    if (o->triangulate_1d) {
      /* ext FeatureInitializer::single_triangulation_1d: the bearing of the anchor observation is taken as exact, the depth
       * along it solves  sum_i |S_i a|^2 d = sum_i (S_i a).(S_i p_CiinA)  over the other observations, S_i = skew(b_i) */
      double a[3] = {(double)uvn[2 * (m - 1)], (double)uvn[2 * (m - 1) + 1], 1.0};
      const double na = sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
      for (int q = 0; q < 3; ++q) a[q] /= na;
      double A1 = 0.0, b1 = 0.0;
      for (int k = 0; k < m - 1; ++k) {
        const tri_rel *r = rel + k;
        const double bc[3] = {(double)uvn[2 * k], (double)uvn[2 * k + 1], 1.0};
        double bi[3], S[9], Sa[3], Sp[3];
        for (int q = 0; q < 3; ++q) bi[q] = r->R_AtoC[q] * bc[0] + r->R_AtoC[3 + q] * bc[1] + r->R_AtoC[6 + q] * bc[2];
        const double nb = sqrt(bi[0] * bi[0] + bi[1] * bi[1] + bi[2] * bi[2]);
        for (int q = 0; q < 3; ++q) bi[q] /= nb;
        skew3(bi, S);
        mat3_vec(S, a, Sa);
        mat3_vec(S, r->p_CinA, Sp);
        A1 += Sa[0] * Sa[0] + Sa[1] * Sa[1] + Sa[2] * Sa[2];
        b1 += Sa[0] * Sp[0] + Sa[1] * Sp[1] + Sa[2] * Sp[2];
      }
      const double depth = b1 / A1;
      for (int q = 0; q < 3; ++q) pA[q] = depth * a[q];
      const double nrm1 = sqrt(pA[0] * pA[0] + pA[1] * pA[1] + pA[2] * pA[2]);
      if (pA[2] < o->min_dist || pA[2] > o->max_dist || isnan(nrm1)) continue;
    } else {
      if (tri_solve3(A, b, pA)) continue;
      tri_sym3_eig(A, ev);
      double emax = fmax(ev[0], fmax(ev[1], ev[2])), emin = fmin(ev[0], fmin(ev[1], ev[2]));
      const double condA = emax / emin;
      const double nrm0 = sqrt(pA[0] * pA[0] + pA[1] * pA[1] + pA[2] * pA[2]);
      if (fabs(condA) > o->max_cond_number || pA[2] < o->min_dist || pA[2] > o->max_dist || isnan(nrm0)) continue;
    }
    if (o->refine_features) {
      double rho = 1.0 / pA[2], alpha = pA[0] / pA[2], beta = pA[1] / pA[2];
      double lam = o->init_lamda, eps = 10000.0;
      int runs = 0, recompute = 1;
      double Hess[9] = {0}, grad[3] = {0};
      double cost_old = tri_cost(rel, uvn, m, alpha, beta, rho);
      while (runs < o->max_runs && lam < o->max_lamda && eps > o->min_dx) {
        if (recompute) {
          memset(Hess, 0, sizeof(Hess));
          memset(grad, 0, sizeof(grad));
          for (int k = 0; k < m; ++k) {
            const double *R = rel[k].R_AtoC, *p = rel[k].p_AinC;
            const double hi1 = R[0] * alpha + R[1] * beta + R[2] + rho * p[0];
            const double hi2 = R[3] * alpha + R[4] * beta + R[5] + rho * p[1];
            const double hi3 = R[6] * alpha + R[7] * beta + R[8] + rho * p[2];
            const double h32 = hi3 * hi3;
            const double H[6] = {(R[0] * hi3 - hi1 * R[6]) / h32, (R[1] * hi3 - hi1 * R[7]) / h32, (p[0] * hi3 - hi1 * p[2]) / h32,
                                 (R[3] * hi3 - hi2 * R[6]) / h32, (R[4] * hi3 - hi2 * R[7]) / h32, (p[1] * hi3 - hi2 * p[2]) / h32};
            const float z0 = (float)(hi1 / hi3), z1 = (float)(hi2 / hi3);
            const double r0 = (double)(uvn[2 * k] - z0), r1 = (double)(uvn[2 * k + 1] - z1);
            for (int i = 0; i < 3; ++i) {
              grad[i] += H[i] * r0 + H[3 + i] * r1;
              for (int j = 0; j < 3; ++j) Hess[3 * i + j] += H[i] * H[j] + H[3 + i] * H[3 + j];
            }
          }
        }
        double Hl[9], dx[3];
        memcpy(Hl, Hess, sizeof(Hl));
        for (int i = 0; i < 3; ++i) Hl[4 * i] *= (1.0 + lam);
        if (tri_solve3(Hl, grad, dx)) break;
        const double cost = tri_cost(rel, uvn, m, alpha + dx[0], beta + dx[1], rho + dx[2]);
        if (cost <= cost_old && (cost_old - cost) / cost_old < o->min_dcost) {
          alpha += dx[0];
          beta += dx[1];
          rho += dx[2];
          eps = 0;
          break;
        }
        if (cost <= cost_old) {
          recompute = 1;
          cost_old = cost;
          alpha += dx[0];
          beta += dx[1];
          rho += dx[2];
          runs++;
          lam = lam / o->lam_mult;
          eps = sqrt(dx[0] * dx[0] + dx[1] * dx[1] + dx[2] * dx[2]);
        } else {
          recompute = 0;
          lam = lam * o->lam_mult;
          continue;
        }
      }
      pA[0] = alpha / rho;
      pA[1] = beta / rho;
      pA[2] = 1.0 / rho;
      /* largest baseline orthogonal to the bearing of the feature */
      const double np = sqrt(pA[0] * pA[0] + pA[1] * pA[1] + pA[2] * pA[2]);
      double base_line_max = 0.0;
      for (int k = 0; k < m; ++k) {
        const double *c = rel[k].p_CinA;
        const double along = (c[0] * pA[0] + c[1] * pA[1] + c[2] * pA[2]) / np;
        const double n2 = c[0] * c[0] + c[1] * c[1] + c[2] * c[2] - along * along;
        const double bl = n2 > 0.0 ? sqrt(n2) : 0.0;
        if (bl > base_line_max) base_line_max = bl;
      }
      if (pA[2] < o->min_dist || pA[2] > o->max_dist || (np / base_line_max) > o->max_baseline || isnan(np)) continue;
    }
    for (int a = 0; a < 3; ++a)
      p_FinG_out[3 * f + a] = R_GtoA[a] * pA[0] + R_GtoA[3 + a] * pA[1] + R_GtoA[6 + a] * pA[2] + p_AinG[a];
    ok[f] = 1;
  }
  free(rel);
  free(Rc);
  free(pc);
  return 0;
}

/* ---------------------------------------------------------------------------------------------
 * update/UpdaterSLAM.cpp:708-850  perform_anchor_change for one landmark held in an anchored representation `rep`
 * (2..5): position re-expressed in the new anchor camera frame (current and first estimates), anchor-change Jacobian
 *   Phi = H_f,new^-1 [ H_x,old | H_x,new terms (subtracted) | H_f,old ]      over  [old anchor, extrinsics, new anchor, landmark]
 * and StateHelper::EKFPropagation of the landmark's block with Q = 0.  P is n x n column-major, lm_id the landmark's id.
 * ------------------------------------------------------------------------------------------- */
int ovo_anchor_change(const ovo_opts *o, const ovo_state *st, int rep, int old_ci, int new_ci, int lm_id,
                      const double p_FinA_old[3], const double p_FinA_old_fej[3], double *P, double p_FinA_new[3],
                      double p_FinA_new_fej[3]) {
  if (rep < 2 || rep > 5) return -1;
  const int n = st->n_state;
  double R_ItoC[9];
  ovo_quat_2_rot(st->calib_q, R_ItoC);
  const double *p_IinC = st->calib_p;
  double p_FinG[3] = {0, 0, 0};
  for (int pass = 0; pass < 2; ++pass) { /* :739-775, pass 0 = current estimates, 1 = first estimates */
    const double *cq = pass ? st->clone_q_fej : st->clone_q, *cp = pass ? st->clone_p_fej : st->clone_p;
    const double *pin = pass ? p_FinA_old_fej : p_FinA_old;
    double *pout = pass ? p_FinA_new_fej : p_FinA_new;
    double Ro[9], Rn[9], R_GtoOLD[9], R_GtoNEW[9], p_OLD[3], p_NEW[3], g[3];
    ovo_quat_2_rot(cq + 4 * old_ci, Ro);
    ovo_quat_2_rot(cq + 4 * new_ci, Rn);
    mat3_mul(R_ItoC, Ro, R_GtoOLD);
    mat3_mul(R_ItoC, Rn, R_GtoNEW);
    for (int i = 0; i < 3; ++i) {
      p_OLD[i] = cp[3 * old_ci + i] - (R_GtoOLD[i] * p_IinC[0] + R_GtoOLD[3 + i] * p_IinC[1] + R_GtoOLD[6 + i] * p_IinC[2]);
      p_NEW[i] = cp[3 * new_ci + i] - (R_GtoNEW[i] * p_IinC[0] + R_GtoNEW[3 + i] * p_IinC[1] + R_GtoNEW[6 + i] * p_IinC[2]);
    }
    for (int i = 0; i < 3; ++i) g[i] = R_GtoOLD[i] * pin[0] + R_GtoOLD[3 + i] * pin[1] + R_GtoOLD[6 + i] * pin[2] + p_OLD[i];
    if (!pass) memcpy(p_FinG, g, sizeof(g));
    for (int i = 0; i < 3; ++i) g[i] -= p_NEW[i];
    mat3_vec(R_GtoNEW, g, pout);
  }
  double Hf_old[9], Hf_new[9], Ha_old[18], Hc_old[18], Ha_new[18], Hc_new[18];
  int nl, nl2;
  if (ovo_feature_jacobian_representation(o, st, rep, p_FinG, old_ci, Hf_old, &nl, Ha_old, Hc_old)) return -2;
  if (ovo_feature_jacobian_representation(o, st, rep, p_FinG, new_ci, Hf_new, &nl2, Ha_new, Hc_new)) return -2;
  /* H_f,new^-1 (3 x 3), or the pseudo-inverse of the 3 x 1 column (:815-819) */
  double Hinv[9];
  if (nl == 1) {
    const double nn = Hf_new[0] * Hf_new[0] + Hf_new[1] * Hf_new[1] + Hf_new[2] * Hf_new[2];
    for (int k = 0; k < 3; ++k) Hinv[k] = Hf_new[k] / nn;
  } else {
    double A[9], Ai[9]; /* small_inverse works column-major */
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) A[3 * j + i] = Hf_new[3 * i + j];
    if (small_inverse(A, 3, Ai)) return -3;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) Hinv[3 * i + j] = Ai[3 * j + i];
  }
  /* order_OLD: old anchor, extrinsics (if estimated), new anchor, landmark (:787-808) */
  int old_id[4], old_size[4], n_old = 0, col_cal = -1, col_new, col_lm;
  int cols = 0;
  old_id[n_old] = st->clone_id[old_ci];
  old_size[n_old++] = 6;
  cols += 6;
  if (o->do_calib_camera_pose) {
    col_cal = cols;
    old_id[n_old] = st->calib_id;
    old_size[n_old++] = 6;
    cols += 6;
  }
  col_new = cols;
  old_id[n_old] = st->clone_id[new_ci];
  old_size[n_old++] = 6;
  cols += 6;
  col_lm = cols;
  old_id[n_old] = lm_id;
  old_size[n_old++] = nl;
  cols += nl;
  double *Phi = (double *)calloc((size_t)nl * cols, sizeof(double));
  double *Q = (double *)calloc((size_t)nl * nl, sizeof(double));
#define OVO_ADD(col0, B, bc, sgn)                                          \
  for (int i = 0; i < nl; ++i)                                             \
    for (int j = 0; j < (bc); ++j) {                                       \
      double a = 0;                                                        \
      for (int k = 0; k < 3; ++k) a += Hinv[3 * i + k] * (B)[(bc)*k + j];  \
      CM(Phi, nl, i, (col0) + j) += (sgn)*a;                               \
    }
  OVO_ADD(0, Ha_old, 6, 1.0);
  if (col_cal >= 0) OVO_ADD(col_cal, Hc_old, 6, 1.0);
  OVO_ADD(col_lm, Hf_old, nl, 1.0);
  OVO_ADD(col_new, Ha_new, 6, -1.0);
  if (col_cal >= 0) OVO_ADD(col_cal, Hc_new, 6, -1.0);
#undef OVO_ADD
  int neg = 0;
  int rc = ovo_ekf_propagation(P, n, lm_id, nl, old_id, old_size, n_old, Phi, Q, &neg);
  free(Phi);
  free(Q);
  return rc ? rc : (neg ? -4 : 0);
}
