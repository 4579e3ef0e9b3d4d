/*
 * TEST INFRASTRUCTURE ONLY  -  see ovp_planefit.c.  CPU restatement of PlaneFitting::{fit_plane, plane_fitting, optimize_plane}
 * (track_plane/PlaneFitting.cpp) of rpng/ov_plane; citations relative to /root/reference/ov_plane/src/.  PARITY UNPINNED.
 */
#ifndef OVP_PLANEFIT_H
#define OVP_PLANEFIT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  uint32_t mt[624];
  int idx;
} ovo_mt19937;

void ovo_mt_seed(ovo_mt19937 *g, uint32_t seed);
uint32_t ovo_mt_next(ovo_mt19937 *g);
/* uniform_int_distribution{0, urange}; variant 0 = libstdc++ of GCC <= 10, 1 = GCC >= 11 */
uint64_t ovo_uniform_int(ovo_mt19937 *g, uint64_t urange, int variant);
void ovo_shuffle(int *v, int n, ovo_mt19937 *g, int variant);

/* pts: n x 3 row-major p_FinG.  Returns 1 on success, abcd = (n, d) with |n| = 1, n.p + d = 0. */
int ovo_fit_plane(const double *pts, int n, double cond_thresh, int cond_check, double abcd[4]);
int ovo_plane_fitting(const double *pts, int n, int min_inlier_num, double max_cond, int variant, double abcd[4],
                      unsigned char *inlier, int *n_inliers);

/* One call of optimize_plane.  Observations are stored feature after feature (obs_start / n_obs); a feature with
 * n_obs = 0 is a SLAM feature (constant, one inflated constraint; PlaneFitting.cpp:274-279). */
typedef struct {
  int n_feats;
  const double *p_FinG;    /* [n_feats*3] */
  const int *obs_start;    /* [n_feats] */
  const int *n_obs;        /* [n_feats] */
  const double *uv_norm;   /* [n_obs_total*2]  Feature::uvs_norm (f32 in the reference, widened) */
  const double *R_GtoC;    /* [n_obs_total*9]  clonesCAM[cam][t].Rot(), row-major */
  const double *p_CinG;    /* [n_obs_total*3]  clonesCAM[cam][t].pos() */
  double cp[3];            /* initial cp_inG */
  double sigma_px_norm, sigma_c;
  int fix_plane;
  double R_GtoI[9], p_IinG[3]; /* stateI  (quat_2_Rot of its first four entries) */
  double R_ItoC[9], p_IinC[3]; /* calib0 */
} ovo_planeopt_problem;

/* returns 1 on success; cp_out / p_out (n_feats*3) / kept (n_feats) as the reference leaves cp_inG, feat->p_FinG and feats */
int ovo_optimize_plane(const ovo_planeopt_problem *pr, double cp_out[3], double *p_out, unsigned char *kept, int *n_kept,
                       int *iterations);
/* robust cost 0.5 sum rho(s) and its gradient at (p_FinG, cp) with every block free (pins) */
double ovo_planeopt_cost(const ovo_planeopt_problem *pr, const double *p_FinG, const double *cp, double *grad_p,
                         double *grad_cp);

#ifdef __cplusplus
}
#endif
#endif
