// UpdaterSLAM::update on the device (update/UpdaterSLAM.cpp:424-673): one workgroup per landmark that has new measurements.
//
//   rows      get_feature_jacobian_full for a landmark that is a state variable (update/UpdaterHelper.cpp:195-513): the bearing rows
//             of its observations over [clone | calibration | landmark] and, when it lies on a plane of the state, the m identical
//             point-on-plane rows over [landmark | closest point] (:448-512) - built here from the device tables for GLOBAL_3D
//             landmarks; landmarks in an anchored / inverse-depth representation arrive with their dense block from the host
//             (the representation Jacobians of :35-193 are host scalar code), the gate below is the same
//   gate      chi2 = res^T (H P_marg H^T + I)^-1 res against the RESIDENT covariance (:526-547): P_marg is gathered column chunk by
//             column chunk into LDS, S eliminated in LDS (Gaussian elimination without pivoting = the LLT of :532, one barrier per
//             column).  The bearing rows come first, so the statistic of the no-plane fallback (:547-609: the same rows without
//             the constraint) is the partial sum over the leading 2 m pivots of the SAME elimination - no second pass
//   scatter   accepted rows go into the stacked system H^T [global columns][rows] that StateHelper::EKFUpdate (k_init.hip S-form,
//             or the information form above 80 rows) reads; rejected landmarks / dropped constraint rows stay zero rows, which
//             change neither the correction nor the covariance (S gets a unit pivot, W a zero column)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "k_slam.h"
#include "ovp_feat_model.h"

namespace ovp {

#define SL_CW 16  // columns of P_marg staged per chunk

__global__ __launch_bounds__(256) void k_slam_gate(SlamParams sp) {
  extern __shared__ double sm[];
  const int l = blockIdx.x, t = threadIdx.x;
  const FeatParams& p = sp.fp;
  const bool pre = sp.pre_rows && sp.pre_rows[l] > 0;
  const int m = pre ? 0 : p.n_meas[l];
  const bool plane = !pre && sp.plane_sid && sp.plane_sid[l] >= 0;
  const int ncal = __popc(p.calmask & 0x3FFFu);
  const int cols = pre ? sp.pre_cols[l] : 6 * m + ncal + 3 + (plane ? 3 : 0);
  const int nb = pre ? sp.pre_rows[l] : 2 * m;            // rows of the fallback (bearing rows)
  const int rows = pre ? nb : (plane ? 3 * m : 2 * m);
  // LDS: S [rows_max][rows_max + 2] (S | res) | r0 [rows_max] | HPc [rows_max][CW] | Pc [cols_max][CW] | ids [cols_max] | H
  const int ldS = sp.rows_max + 2;
  double* S = sm;
  double* r0 = S + (size_t)sp.rows_max * ldS;
  double* HPc = r0 + sp.rows_max;
  double* Pc = HPc + (size_t)sp.rows_max * SL_CW;
  int* ids = (int*)(Pc + (size_t)sp.cols_max * SL_CW);
  double* Hl = (double*)(ids + ((sp.cols_max + 1) & ~1));
  double* H = sp.h_in_lds ? Hl : sp.Hscr + (size_t)l * sp.rows_max * sp.cols_max;  // [rows][cols] row-major
  __shared__ int bad_b, bad_f;
  if (t == 0) bad_b = 0, bad_f = 0;
  if (rows < 1) {
    if (t == 0) sp.status[l] = 0, sp.chi2[l] = 0.0;
    return;
  }
  for (int e = t; e < rows * cols; e += 256) H[e] = 0.0;
  for (int e = t; e < rows * ldS; e += 256) {
    const int i = e / ldS, j = e - i * ldS;
    S[e] = (i == j) ? 1.0 : 0.0;  // R = I (:531)
  }
  if (pre) {
    const double* blk = sp.pre_H + sp.pre_off[l];  // [rows x cols] column-major, then res [rows]
    const int* pid = sp.pre_ids + sp.pre_ids_off[l];
    for (int e = t; e < cols; e += 256) ids[e] = pid[e];
    __syncthreads();
    for (int e = t; e < rows * cols; e += 256) {
      const int k = e / rows, i = e - k * rows;
      H[(size_t)i * cols + k] = blk[e];
    }
    for (int e = t; e < rows; e += 256) r0[e] = blk[(size_t)rows * cols + e];
  } else {
    // local column order: [clone blocks in observation order | estimated calibration columns | landmark | closest point]
    for (int e = t; e < 6 * m; e += 256) {
      const int a = e / 6;
      ids[e] = p.clone_id[p.clone_idx[(size_t)l * p.max_meas + a]] + (e - 6 * a);
    }
    if (t < 14 && ((p.calmask >> t) & 1)) ids[6 * m + __popc(p.calmask & ((1u << t) - 1u))] = p.calcol[t];
    if (t < 3) ids[6 * m + ncal + t] = sp.lm_id[l] + t;
    if (plane && t < 3) ids[6 * m + ncal + 3 + t] = sp.plane_sid[l] + t;
    __syncthreads();  // (H zeroed)
    if (t < 2 * m) {
      const int a = t >> 1, r = t & 1;
      const int ci = p.clone_idx[(size_t)l * p.max_meas + a];
      double jrow[6], crow[14], hf[3], res;
      build_bearing_row<true>(p, l, a, r, true, ci, jrow, crow, hf, res, sp.p_fej + 3 * l);
      double* h = H + (size_t)t * cols;
#pragma unroll
      for (int k = 0; k < 6; ++k) h[6 * a + k] = jrow[k];
#pragma unroll
      for (int k = 0; k < 14; ++k)
        if ((p.calmask >> k) & 1) h[6 * m + __popc(p.calmask & ((1u << k) - 1u))] = crow[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) h[6 * m + ncal + k] = hf[k];
      r0[t] = res;
    }
    if (plane && t >= 64 && t < 64 + m) {
      // point-on-plane row (update/UpdaterHelper.cpp:448-512), once per observation (:503-511); the plane is a state variable
      const int a = t - 64;
      const double* pv = p.p_FinG + 3 * l;
      const double* cpv = sp.cp + 3 * l;
      double d = sqrt(cpv[0] * cpv[0] + cpv[1] * cpv[1] + cpv[2] * cpv[2]);
      double n0 = cpv[0] / d, n1 = cpv[1] / d, n2 = cpv[2] / d;
      const double res = sp.white_c * (0.0 - (n0 * pv[0] + n1 * pv[1] + n2 * pv[2] - d));
      double q0 = pv[0], q1 = pv[1], q2 = pv[2];
      if (p.do_fej) {  // :467-476
        const double* pf = sp.p_fej + 3 * l;
        const double* cf = sp.cp_fej + 3 * l;
        q0 = pf[0], q1 = pf[1], q2 = pf[2];
        d = sqrt(cf[0] * cf[0] + cf[1] * cf[1] + cf[2] * cf[2]);
        n0 = cf[0] / d, n1 = cf[1] / d, n2 = cf[2] / d;
      }
      const double np = n0 * q0 + n1 * q1 + n2 * q2;
      const double s = sp.white_c * 1.0 / d;
      double* h = H + (size_t)(2 * m + a) * cols;
      h[6 * m + ncal + 0] = sp.white_c * n0;  // H_f row (:497)
      h[6 * m + ncal + 1] = sp.white_c * n1;
      h[6 * m + ncal + 2] = sp.white_c * n2;
      h[6 * m + ncal + 3] = s * (q0 - np * n0 - d * n0);  // H_c_plane (:479-481)
      h[6 * m + ncal + 4] = s * (q1 - np * n1 - d * n1);
      h[6 * m + ncal + 5] = s * (q2 - np * n2 - d * n2);
      r0[2 * m + a] = res;
    }
  }
  __syncthreads();
  for (int e = t; e < rows; e += 256) S[(size_t)e * ldS + rows] = r0[e];
  // S += H P_marg H^T, chunk of SL_CW marginal columns at a time
  for (int j0 = 0; j0 < cols; j0 += SL_CW) {
    const int cw = min(SL_CW, cols - j0);
    for (int e = t; e < cols * SL_CW; e += 256) {
      const int k = e / SL_CW, jj = e - k * SL_CW;
      Pc[e] = jj < cw ? p.P[(size_t)ids[k] * p.ldp + ids[j0 + jj]] : 0.0;
    }
    __syncthreads();
    for (int e = t; e < rows * SL_CW; e += 256) {
      const int i = e / SL_CW, jj = e - i * SL_CW;
      const double* h = H + (size_t)i * cols;
      double s0 = 0.0, s1 = 0.0;
      int k = 0;
      for (; k + 1 < cols; k += 2) {
        s0 = fma(h[k], Pc[k * SL_CW + jj], s0);
        s1 = fma(h[k + 1], Pc[(k + 1) * SL_CW + jj], s1);
      }
      if (k < cols) s0 = fma(h[k], Pc[k * SL_CW + jj], s0);
      HPc[e] = s0 + s1;
    }
    __syncthreads();
    for (int e = t; e < rows * rows; e += 256) {
      const int i = e / rows, i2 = e - i * rows;
      const double* hp = HPc + (size_t)i * SL_CW;
      const double* h2 = H + (size_t)i2 * cols + j0;
      double s = 0.0;
      for (int jj = 0; jj < cw; ++jj) s = fma(hp[jj], h2[jj], s);
      S[(size_t)i * ldS + i2] += s;
    }
    __syncthreads();
  }
  // elimination of [S | res]: after step c - 1 row c is final, y_c^2 = res_c'^2 / pivot_c
  const int W = rows + 1;
  for (int c = 0; c < rows; ++c) {
    const double* rc = S + (size_t)c * ldS;
    const double piv = rc[c];
    if (t == 0 && !(piv > 0.0)) {
      bad_f = 1;
      if (c < nb) bad_b = 1;
    }
    const double ip = 1.0 / piv;
    for (int e = t; e < (rows - c - 1) * (W - c - 1); e += 256) {
      const int i = c + 1 + e / (W - c - 1), j = c + 1 + (e - (i - c - 1) * (W - c - 1));
      S[(size_t)i * ldS + j] = fma(-(S[(size_t)i * ldS + c] * ip), rc[j], S[(size_t)i * ldS + j]);
    }
    __syncthreads();
  }
  __shared__ int st_sh;
  if (t == 0) {
    double chi2_b = 0.0, chi2_f = 0.0;
    for (int c = 0; c < rows; ++c) {
      const double y = S[(size_t)c * ldS + rows];
      const double piv = S[(size_t)c * ldS + c];
      const double v = y * y / (piv > 0.0 ? piv : 1.0);
      chi2_f += v;
      if (c < nb) chi2_b += v;
    }
    if (bad_b) chi2_b = 1e300;
    if (bad_f) chi2_f = 1e300;
    const double thr_f = p.chi2_mult * p.chi2_table[rows < OVP_CHI2_TABLE ? rows : OVP_CHI2_TABLE];
    const double thr_b = p.chi2_mult * p.chi2_table[nb < OVP_CHI2_TABLE ? nb : OVP_CHI2_TABLE];
    int st;
    double chi2;
    if (rows > nb && !(chi2_f > thr_f)) st = 1, chi2 = chi2_f;           // with its plane (:541-547 passes)
    else if (rows > nb) st = (chi2_b > thr_b) ? 0 : 2, chi2 = chi2_b;    // no-plane fallback (:547-609)
    else st = (chi2_b > thr_b) ? 0 : 1, chi2 = chi2_b;                   // (:611-620)
    sp.status[l] = (unsigned char)st;
    sp.chi2[l] = chi2;
    st_sh = st;
  }
  __syncthreads();
  const int keep = st_sh == 1 ? rows : (st_sh == 2 ? nb : 0);
  const int row0 = sp.row0[l];
  // the landmark owns its rows of the stacked system: zero them, then drop the accepted rows in (no memset in front of the launch)
  for (int e = t; e < sp.gcols * rows; e += 256) {
    const int g = e / rows, i = e - g * rows;
    sp.Ht[(size_t)g * sp.m_total + row0 + i] = 0.0;
  }
  for (int e = t; e < rows; e += 256) sp.res_out[row0 + e] = e < keep ? r0[e] : 0.0;
  __syncthreads();
  for (int e = t; e < keep * cols; e += 256) {
    const int i = e / cols, k = e - i * cols;
    const double v = H[e];
    if (v != 0.0) sp.Ht[(size_t)sp.gpos[ids[k]] * sp.m_total + row0 + i] = v;
  }
  // its columns of M = P H^T (the S-form update's first product, k_init_m): P[:, ids] H_l^T, zero for rows that were dropped
  if (sp.Mall) {
    // thread <-> state row r, four stacked rows at a time: P[r][ids[k]] is read as P[ids[k]][r] (symmetric) so that the threads of a
    // wave walk one row of P together
    const int n = p.n;
    for (int i0 = 0; i0 < rows; i0 += 4) {
      for (int r = t; r < n; r += 256) {
        double s[4] = {0.0, 0.0, 0.0, 0.0};
        if (i0 < keep) {
          const double* h = H + (size_t)i0 * cols;
          const int ni = min(4, keep - i0);
#pragma unroll 8
          for (int k = 0; k < cols; ++k) {
            const double pv = p.P[(size_t)ids[k] * p.ldp + r];
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (q < ni) s[q] = fma(pv, h[(size_t)q * cols + k], s[q]);
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (i0 + q < rows) sp.Mall[(size_t)r * sp.m_total + row0 + i0 + q] = s[q];
      }
    }
  }
}

}  // namespace ovp

extern "C" {
size_t ovp_slam_gate_lds(int rows_max, int cols_max, int with_h) {
  size_t d = (size_t)rows_max * (rows_max + 2) + rows_max + (size_t)rows_max * SL_CW + (size_t)cols_max * SL_CW +
             (size_t)((cols_max + 1) / 2 + 1);
  if (with_h) d += (size_t)rows_max * cols_max;
  return d * sizeof(double);
}

hipError_t ovp_launch_slam_gate(const ovp::SlamParams* sp, int n_landmarks, size_t lds, hipStream_t stream) {
  static unsigned long long attr_mask = 0;  // per device (ovp_kernels.h)
  if (ovp_lds_attr_needed(&attr_mask)) {
    (void)hipFuncSetAttribute((const void*)ovp::k_slam_gate, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
    (void)hipGetLastError();  // (a kernel with static LDS refuses the full 160 KB: harmless, a real shortage fails the launch itself)
    ovp_lds_attr_done(&attr_mask);
  }
  hipLaunchKernelGGL(ovp::k_slam_gate, dim3(n_landmarks), dim3(256), lds, stream, *sp);
  return hipGetLastError();
}
}
