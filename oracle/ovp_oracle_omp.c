/* TEST / BENCH INFRASTRUCTURE - never part of the product path.
 *
 * All-cores CPU ceiling of the MSCKF point update ("CPU-omp"): the same algorithm as
 * ovo_msckf_point_update (oracle/ovp_oracle.c, which restates update/UpdaterMSCKF.cpp:671-814 in the reference's own loop order
 * on one thread), with the two stages the reference runs sequentially spread over the host cores:
 *   - the per-feature stage (Jacobian, Givens nullspace projection, chi2 gate; UpdaterMSCKF.cpp:695-786) is an OpenMP loop over
 *     the features - they are independent until the stack;
 *   - the measurement compression (UpdaterHelper.cpp:548-579) becomes a two-level TSQR: every thread reduces a slab of the stack
 *     to its c x c triangle with Householder reflections, the triangles are reduced in groups, then once more.  R^T R and R^T Q^T r
 *     are what the update depends on, so the result equals the sequential Givens sweep's to rounding.
 * The EKF update itself (StateHelper.cpp:121-202) is small and stays on one thread.  The plane loop is NOT parallelised here: it is
 * sequential across planes by definition and the rows its Givens sweep retains decide its chi2.
 * The reference ships no such path (it runs on one thread, ov_plane/CMakeLists.txt:22): this is context for the speed-up figure,
 * not a parity oracle - tests pin it against the sequential restatement. */
#include "ovp_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define CM(A, ld, r, c) ((A)[(size_t)(c) * (size_t)(ld) + (size_t)(r)])

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* Householder QR of A (rows x cols, col-major, ld) in place, applied to rhs as well; on return the upper triangle of the first
 * min(rows, cols) rows holds R and rhs[0..] holds Q^T rhs. */
static void householder_qr(double *A, int rows, int cols, int ld, double *rhs) {
  const int kmax = rows < cols ? rows : cols;
  for (int k = 0; k < kmax; ++k) {
    double *v = &CM(A, ld, k, k);
    const int len = rows - k;
    double nrm2 = 0.0;
    for (int i = 0; i < len; ++i) nrm2 += v[i] * v[i];
    if (nrm2 == 0.0) continue;
    const double nrm = sqrt(nrm2);
    const double alpha = v[0] > 0 ? -nrm : nrm;
    const double v0 = v[0] - alpha;
    /* H = I - beta u u^T with u = [v0, v[1..]] ; beta = 1 / (nrm2 - alpha * v[0]) */
    const double denom = nrm2 - alpha * v[0];
    if (denom == 0.0) continue;
    const double beta = 1.0 / denom;
    v[0] = v0;
    for (int j = k + 1; j < cols; ++j) {
      double *a = &CM(A, ld, k, j);
      double dot = 0.0;
      for (int i = 0; i < len; ++i) dot += v[i] * a[i];
      const double t = beta * dot;
      for (int i = 0; i < len; ++i) a[i] -= t * v[i];
    }
    {
      double dot = 0.0;
      for (int i = 0; i < len; ++i) dot += v[i] * rhs[k + i];
      const double t = beta * dot;
      for (int i = 0; i < len; ++i) rhs[k + i] -= t * v[i];
    }
    v[0] = alpha;
    for (int i = 1; i < len; ++i) v[i] = 0.0;
  }
}

/* reduces slabs [r0, r1) of the col-major stack (ld) to triangles: out (nb * cols rows x cols, col-major, ld = nb * cols) */
static void tsqr_level(const double *H, const double *res, size_t ld, const size_t *r0, const size_t *r1, int nb, int cols,
                       double *outH, double *outr) {
  const size_t ldo = (size_t)nb * cols;
#pragma omp parallel for schedule(dynamic, 1)
  for (int b = 0; b < nb; ++b) {
    const size_t rows = r1[b] - r0[b];
    const size_t rl = rows > (size_t)cols ? rows : (size_t)cols;
    double *A = (double *)calloc(rl * (size_t)cols, sizeof(double));
    double *y = (double *)calloc(rl, sizeof(double));
    for (int j = 0; j < cols; ++j) memcpy(A + (size_t)j * rl, H + (size_t)j * ld + r0[b], sizeof(double) * rows);
    memcpy(y, res + r0[b], sizeof(double) * rows);
    householder_qr(A, (int)rl, cols, (int)rl, y);
    for (int j = 0; j < cols; ++j)
      for (int i = 0; i < cols; ++i) outH[(size_t)j * ldo + (size_t)b * cols + i] = (i <= j) ? A[(size_t)j * rl + i] : 0.0;
    for (int i = 0; i < cols; ++i) outr[(size_t)b * cols + i] = y[i];
    free(A);
    free(y);
  }
}

int ovo_msckf_point_update_omp(const ovo_opts *o, const ovo_state *st, const ovo_feats *fb, double *P, double *dx,
                               uint8_t *accepted, double *chi2_out, double *timings, int n_threads) {
#ifdef _OPENMP
  if (n_threads > 0) omp_set_num_threads(n_threads);
  const int T = omp_get_max_threads();
#else
  const int T = 1;
  (void)n_threads;
#endif
  const double t0 = now_s();
  const int n = st->n_state;
  const int F = fb->n_feats;
  const int mm = fb->max_meas;
  /* fixed column order of the stack: calibration (UpdaterHelper.cpp:205-277 puts it first), then the clones by slot */
  int *map_col = (int *)malloc(sizeof(int) * (size_t)n);
  for (int i = 0; i < n; ++i) map_col[i] = -1;
  int *order_id = (int *)malloc(sizeof(int) * (size_t)(st->n_clones + 4));
  int *order_size = (int *)malloc(sizeof(int) * (size_t)(st->n_clones + 4));
  int n_order = 0, cols_big = 0;
  if (o->do_calib_camera_pose) {
    map_col[st->calib_id] = cols_big;
    order_id[n_order] = st->calib_id;
    order_size[n_order++] = 6;
    cols_big += 6;
  }
  if (o->do_calib_camera_intrinsics) {
    map_col[st->intr_id] = cols_big;
    order_id[n_order] = st->intr_id;
    order_size[n_order++] = 8;
    cols_big += 8;
  }
  for (int c = 0; c < st->n_clones; ++c) {
    map_col[st->clone_id[c]] = cols_big;
    order_id[n_order] = st->clone_id[c];
    order_size[n_order++] = 6;
    cols_big += 6;
  }
  /* row offsets: every feature owns its 2m - 3 rows whether it passes the gate or not (rejected ones stay zero rows) */
  size_t *row0 = (size_t *)malloc(sizeof(size_t) * (size_t)(F + 1));
  row0[0] = 0;
  for (int f = 0; f < F; ++f) {
    const int m = fb->n_meas[f];
    row0[f + 1] = row0[f] + (size_t)(m >= 2 ? 2 * m - 3 : 0);
  }
  const size_t rows_big = row0[F];
  double *Hbig = (double *)calloc((rows_big ? rows_big : 1) * (size_t)cols_big, sizeof(double));
  double *rbig = (double *)calloc(rows_big ? rows_big : 1, sizeof(double));

#pragma omp parallel
  {
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
#pragma omp for schedule(dynamic, 4)
    for (int f = 0; f < F; ++f) {
      accepted[f] = 0;
      chi2_out[f] = 0.0;
      if (fb->n_meas[f] < 2) continue;
      int rows, cols, hfc, no;
      ovo_feature_jacobian_full(o, st, fb, f, o->sigma_constraint, 0, NULL, NULL, -1, H_f, H_x, res, &rows, &cols, &hfc, oid, osz,
                                &no);
      ovo_nullspace_project(H_f, rows, hfc, H_x, cols, NULL, 0, res);
      const int q = rows - hfc;
      ovo_marginal_cov(P, n, oid, osz, no, Pm);
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
      /* forward / backward substitution with the factor (same arithmetic as the sequential restatement) */
      for (int i = 0; i < q; ++i) {
        double s = tmp[i];
        for (int k = 0; k < i; ++k) s -= CM(S, q, i, k) * tmp[k];
        tmp[i] = s / CM(S, q, i, i);
      }
      for (int i = q - 1; i >= 0; --i) {
        double s = tmp[i];
        for (int k = i + 1; k < q; ++k) s -= CM(S, q, k, i) * tmp[k];
        tmp[i] = s / CM(S, q, i, i);
      }
      double chi2 = 0.0;
      for (int i = 0; i < q; ++i) chi2 += res[hfc + i] * tmp[i];
      chi2_out[f] = chi2;
      if (chi2 > o->chi2_multiplier * ovo_chi2_quantile_095(q)) continue;
      accepted[f] = 1;
      int ct_hx = 0;
      for (int v = 0; v < no; ++v) {
        const int c0 = map_col[oid[v]];
        for (int cc = 0; cc < osz[v]; ++cc)
          for (int i = 0; i < q; ++i) CM(Hbig, rows_big, row0[f] + i, c0 + cc) = CM(H_x, rows, hfc + i, ct_hx + cc);
        ct_hx += osz[v];
      }
      for (int i = 0; i < q; ++i) rbig[row0[f] + i] = res[hfc + i];
    }
    free(H_f);
    free(H_x);
    free(res);
    free(Pm);
    free(HP);
    free(S);
    free(tmp);
    free(oid);
    free(osz);
  }
  const double t1 = now_s();
  for (int i = 0; i < n; ++i) dx[i] = 0.0;
  int rows_c = 0;
  double t2 = t1, t3 = t1;
  int any = 0;
  for (int f = 0; f < F; ++f) any |= accepted[f];
  if (any && rows_big > 0) {
    /* level 1: one slab per thread (at least 4 c rows each), level 2: groups of eight triangles, level 3: the rest */
    int nb = T;
    while (nb > 1 && rows_big / (size_t)nb < (size_t)(4 * cols_big)) --nb;
    size_t *r0 = (size_t *)malloc(sizeof(size_t) * (size_t)nb), *r1 = (size_t *)malloc(sizeof(size_t) * (size_t)nb);
    for (int b = 0; b < nb; ++b) {
      r0[b] = rows_big * (size_t)b / (size_t)nb;
      r1[b] = rows_big * (size_t)(b + 1) / (size_t)nb;
    }
    double *H1 = (double *)malloc(sizeof(double) * (size_t)nb * cols_big * cols_big);
    double *y1 = (double *)malloc(sizeof(double) * (size_t)nb * cols_big);
    tsqr_level(Hbig, rbig, rows_big, r0, r1, nb, cols_big, H1, y1);
    free(r0);
    free(r1);
    int cur_nb = nb;
    double *Hc = H1, *yc = y1;
    while (cur_nb > 1) {
      const int ng = (cur_nb + 7) / 8;
      size_t *g0 = (size_t *)malloc(sizeof(size_t) * (size_t)ng), *g1 = (size_t *)malloc(sizeof(size_t) * (size_t)ng);
      for (int g = 0; g < ng; ++g) {
        g0[g] = (size_t)g * 8 * cols_big;
        g1[g] = (size_t)((g + 1) * 8 < cur_nb ? (g + 1) * 8 : cur_nb) * cols_big;
      }
      double *Hn = (double *)malloc(sizeof(double) * (size_t)ng * cols_big * cols_big);
      double *yn = (double *)malloc(sizeof(double) * (size_t)ng * cols_big);
      tsqr_level(Hc, yc, (size_t)cur_nb * cols_big, g0, g1, ng, cols_big, Hn, yn);
      free(g0);
      free(g1);
      free(Hc);
      free(yc);
      Hc = Hn;
      yc = yn;
      cur_nb = ng;
    }
    rows_c = rows_big < (size_t)cols_big ? (int)rows_big : cols_big;
    t2 = now_s();
    int neg = 0;
    ovo_ekf_update(P, n, order_id, order_size, n_order, Hc, rows_c, cols_big, yc, dx, &neg);
    t3 = now_s();
    free(Hc);
    free(yc);
    if (neg) rows_c = -2;
  }
  if (timings) {
    timings[0] = t1 - t0;
    timings[1] = t2 - t1;
    timings[2] = t3 - t2;
    timings[3] = t3 - t0;
  }
  free(map_col);
  free(order_id);
  free(order_size);
  free(row0);
  free(Hbig);
  free(rbig);
  return rows_c >= 0 ? T : rows_c;
}
