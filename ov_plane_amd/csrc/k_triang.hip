// Feature triangulation on the device: ext ov_core::FeatureInitializer::single_triangulation + single_gaussnewton as used by
// update/UpdaterMSCKF.cpp:120-166 (the per-feature step right in front of the update).
// One wavefront per feature, lane k <-> observation k.  Everything a lane contributes to a sum (the 3x3 normal matrix, the
// Levenberg-Marquardt Hessian / gradient, the cost) is staged in LDS and added up by every lane in observation order - the
// reference's own order - so the accept / reject decisions of the damped iteration are the ones the sequential code takes;
// floating-point contraction is switched off for this file for the same reason.  The residuals of the refinement are formed
// in single precision like the reference's (Eigen::Matrix<float,2,1>).
#pragma clang fp contract(off)
#include "ovplane_hip.h"
#include "ovp_dev.h"
#include "ovp_kernels.h"

namespace ovp {

static constexpr int TRI_WAVES = 4;   // features per block
static constexpr int TRI_PITCH = 10;  // doubles of contribution per observation

__device__ __forceinline__ void tri_sym3_eig(const double (&A)[9], double (&ev)[3]) {
  double a[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) a[i] = A[i];
  for (int sweep = 0; sweep < 30; ++sweep) {
    const double off = fabs(a[1]) + fabs(a[2]) + fabs(a[5]);
    // converged: the off-diagonal part is below the rounding of the diagonal (same test as the oracle's; cyclic Jacobi needs four or
    // five sweeps - until round 6 all thirty ran, 90 rotations of five dependent divisions / square roots each: a third of the kernel)
    if (off <= 1e-17 * (fabs(a[0]) + fabs(a[4]) + fabs(a[8]))) break;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int q = p + 1; q < 3; ++q) {
        const double apq = a[3 * p + q];
        if (apq == 0.0) continue;
        const double theta = (a[3 * q + q] - a[3 * p + p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const double akp = a[3 * k + p], akq = a[3 * k + q];
          a[3 * k + p] = c * akp - s * akq;
          a[3 * k + q] = s * akp + c * akq;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
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

// 3x3 solve, Gaussian elimination with full pivoting (identical operation order to the oracle's tri_solve3)
__device__ __forceinline__ bool tri_solve3(const double (&A)[9], const double (&b)[3], double (&x)[3]) {
  double m[12];
  int perm[3] = {0, 1, 2};
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
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
    if (!(best > 0.0)) return false;
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
  return true;
}

// sum of NV staged values per observation, in observation order; every lane gets the same result
template <int NV>
__device__ __forceinline__ void ordered_sum(const double* buf, int m, double (&acc)[NV]) {
#pragma unroll
  for (int j = 0; j < NV; ++j) acc[j] = 0.0;
  for (int k = 0; k < m; ++k) {
#pragma unroll
    for (int j = 0; j < NV; ++j) acc[j] += buf[k * TRI_PITCH + j];
  }
}

__global__ __launch_bounds__(TRI_WAVES * 64) void k_triangulate(const TriParams p) {
  __shared__ __attribute__((aligned(16))) double stage[TRI_WAVES][OVP_MAX_MEAS * TRI_PITCH];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int f = blockIdx.x * TRI_WAVES + wave;
  if (f >= p.n_feats) return;  // whole wave
  double* buf = stage[wave];
  const int m = p.n_meas[f];
  if (m < 2 || m > OVP_MAX_MEAS) {
    if (lane == 0) {
      p.ok[f] = 0;
      p.p_FinG[3 * f] = p.p_FinG[3 * f + 1] = p.p_FinG[3 * f + 2] = 0.0;
    }
    return;
  }
  const bool act = lane < m;
  const int* cidx = p.clone_idx + (size_t)f * p.max_meas;
  const double* cal = p.cal;  // [0..8] R_ItoC, [9..11] p_IinC
  // camera pose of a clone slot: R_GtoC = R_ItoC R_GtoI ; p_CinG = p_IinG - R_GtoC^T p_IinC   (UpdaterMSCKF.cpp:129-130)
  auto cam_pose = [&](int slot, double (&R)[9], double (&pc)[3]) {
    const double* RI = p.clone_R + 9 * slot;
    const double* pI = p.clone_p + 3 * slot;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) R[3 * i + j] = cal[3 * i] * RI[j] + cal[3 * i + 1] * RI[3 + j] + cal[3 * i + 2] * RI[6 + j];
#pragma unroll
    for (int a = 0; a < 3; ++a) pc[a] = pI[a] - (R[a] * cal[9] + R[3 + a] * cal[10] + R[6 + a] * cal[11]);
  };
  double R_GtoA[9], p_AinG[3];
  cam_pose(cidx[m - 1], R_GtoA, p_AinG);  // anchor = last measurement
  double R_GtoCi[9], p_CiinG[3];
  cam_pose(cidx[act ? lane : 0], R_GtoCi, p_CiinG);
  const float un = p.uvn[((size_t)f * p.max_meas + (act ? lane : 0)) * 2];
  const float vn = p.uvn[((size_t)f * p.max_meas + (act ? lane : 0)) * 2 + 1];
  // relative pose of this observation's camera with respect to the anchor
  double R[9], pCinA[3], pAinC[3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      R[3 * i + j] = R_GtoCi[3 * i] * R_GtoA[3 * j] + R_GtoCi[3 * i + 1] * R_GtoA[3 * j + 1] + R_GtoCi[3 * i + 2] * R_GtoA[3 * j + 2];
  {
    const double d[3] = {p_CiinG[0] - p_AinG[0], p_CiinG[1] - p_AinG[1], p_CiinG[2] - p_AinG[2]};
#pragma unroll
    for (int a = 0; a < 3; ++a) pCinA[a] = R_GtoA[3 * a] * d[0] + R_GtoA[3 * a + 1] * d[1] + R_GtoA[3 * a + 2] * d[2];
#pragma unroll
    for (int a = 0; a < 3; ++a) pAinC[a] = -(R[3 * a] * pCinA[0] + R[3 * a + 1] * pCinA[1] + R[3 * a + 2] * pCinA[2]);
  }
  // ---- linear triangulation: A = sum (I - b b^T) formed as S^T S, S = skew(b) ----
  if (act) {
    const double bc[3] = {(double)un, (double)vn, 1.0};
    double bi[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) bi[a] = R[a] * bc[0] + R[3 + a] * bc[1] + R[6 + a] * bc[2];
    const double nb = sqrt(bi[0] * bi[0] + bi[1] * bi[1] + bi[2] * bi[2]);
#pragma unroll
    for (int a = 0; a < 3; ++a) bi[a] /= nb;
    const double S[9] = {0.0, -bi[2], bi[1], bi[2], 0.0, -bi[0], -bi[1], bi[0], 0.0};
    double Ai[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) Ai[3 * i + j] = S[i] * S[j] + S[3 + i] * S[3 + j] + S[6 + i] * S[6 + j];
    double* o = buf + lane * TRI_PITCH;
    o[0] = Ai[0];
    o[1] = Ai[1];
    o[2] = Ai[2];
    o[3] = Ai[4];
    o[4] = Ai[5];
    o[5] = Ai[8];
#pragma unroll
    for (int i = 0; i < 3; ++i) o[6 + i] = Ai[3 * i] * pCinA[0] + Ai[3 * i + 1] * pCinA[1] + Ai[3 * i + 2] * pCinA[2];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  double s9[9];
  ordered_sum<9>(buf, m, s9);
  const double A[9] = {s9[0], s9[1], s9[2], s9[1], s9[3], s9[4], s9[2], s9[4], s9[5]};
  const double b[3] = {s9[6], s9[7], s9[8]};
  double pA[3] = {0.0, 0.0, 0.0}, ev[3];
  bool good;
  if (p.triangulate_1d) {
    // ext single_triangulation_1d: depth d along the anchor bearing a from  sum_i |S_i a|^2 d = sum_i (S_i a).(S_i p_CiinA),
    // S_i = skew(b_i); the anchor's own observation is skipped, no condition-number test
    const float ua = p.uvn[((size_t)f * p.max_meas + (m - 1)) * 2], va = p.uvn[((size_t)f * p.max_meas + (m - 1)) * 2 + 1];
    double a[3] = {(double)ua, (double)va, 1.0};
    const double na = sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
#pragma unroll
    for (int k = 0; k < 3; ++k) a[k] /= na;
    __builtin_amdgcn_wave_barrier();
    if (act) {
      const double bc[3] = {(double)un, (double)vn, 1.0};
      double bi[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) bi[k] = R[k] * bc[0] + R[3 + k] * bc[1] + R[6 + k] * bc[2];
      const double nb = sqrt(bi[0] * bi[0] + bi[1] * bi[1] + bi[2] * bi[2]);
#pragma unroll
      for (int k = 0; k < 3; ++k) bi[k] /= nb;
      const double Sa[3] = {bi[1] * a[2] - bi[2] * a[1], bi[2] * a[0] - bi[0] * a[2], bi[0] * a[1] - bi[1] * a[0]};
      const double Sp[3] = {bi[1] * pCinA[2] - bi[2] * pCinA[1], bi[2] * pCinA[0] - bi[0] * pCinA[2],
                            bi[0] * pCinA[1] - bi[1] * pCinA[0]};
      const bool anchor = lane == m - 1;
      buf[lane * TRI_PITCH + 0] = anchor ? 0.0 : (Sa[0] * Sa[0] + Sa[1] * Sa[1] + Sa[2] * Sa[2]);
      buf[lane * TRI_PITCH + 1] = anchor ? 0.0 : (Sa[0] * Sp[0] + Sa[1] * Sp[1] + Sa[2] * Sp[2]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    double A1 = 0.0, b1 = 0.0;
    for (int k = 0; k < m - 1; ++k) {
      A1 += buf[k * TRI_PITCH + 0];
      b1 += buf[k * TRI_PITCH + 1];
    }
    const double depth = b1 / A1;
#pragma unroll
    for (int k = 0; k < 3; ++k) pA[k] = depth * a[k];
    const double nrm0 = sqrt(pA[0] * pA[0] + pA[1] * pA[1] + pA[2] * pA[2]);
    good = !(pA[2] < p.min_dist || pA[2] > p.max_dist || isnan(nrm0));
  } else {
    good = tri_solve3(A, b, pA);
  }
  if (good && !p.triangulate_1d) {
    tri_sym3_eig(A, ev);
    const double emax = fmax(ev[0], fmax(ev[1], ev[2])), emin = fmin(ev[0], fmin(ev[1], ev[2]));
    const double condA = emax / emin;
    const double nrm0 = sqrt(pA[0] * pA[0] + pA[1] * pA[1] + pA[2] * pA[2]);
    if (fabs(condA) > p.max_cond_number || pA[2] < p.min_dist || pA[2] > p.max_dist || isnan(nrm0)) good = false;
  }
  // cost of a candidate (alpha, beta, rho): residuals in single precision
  auto cost_of = [&](double alpha, double beta, double rho) {
    __builtin_amdgcn_wave_barrier();
    if (act) {
      const double hi1 = R[0] * alpha + R[1] * beta + R[2] + rho * pAinC[0];
      const double hi2 = R[3] * alpha + R[4] * beta + R[5] + rho * pAinC[1];
      const double hi3 = R[6] * alpha + R[7] * beta + R[8] + rho * pAinC[2];
      const float z0 = (float)(hi1 / hi3), z1 = (float)(hi2 / hi3);
      const float r0 = un - z0, r1 = vn - z1;
      const float nrm = sqrtf(r0 * r0 + r1 * r1);
      buf[lane * TRI_PITCH + 9] = (double)nrm * (double)nrm;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    double err = 0.0;
    for (int k = 0; k < m; ++k) err += buf[k * TRI_PITCH + 9];
    return err;
  };
  if (good && p.refine_features) {
    double rho = 1.0 / pA[2], alpha = pA[0] / pA[2], beta = pA[1] / pA[2];
    double lam = p.init_lamda, eps = 10000.0;
    int runs = 0;
    bool recompute = true;
    double Hess[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, grad[3] = {0, 0, 0};
    double cost_old = cost_of(alpha, beta, rho);
    while (runs < p.max_runs && lam < p.max_lamda && eps > p.min_dx) {
      if (recompute) {
        __builtin_amdgcn_wave_barrier();
        if (act) {
          const double hi1 = R[0] * alpha + R[1] * beta + R[2] + rho * pAinC[0];
          const double hi2 = R[3] * alpha + R[4] * beta + R[5] + rho * pAinC[1];
          const double hi3 = R[6] * alpha + R[7] * beta + R[8] + rho * pAinC[2];
          const double h32 = hi3 * hi3;
          const double H[6] = {(R[0] * hi3 - hi1 * R[6]) / h32, (R[1] * hi3 - hi1 * R[7]) / h32, (pAinC[0] * hi3 - hi1 * pAinC[2]) / h32,
                               (R[3] * hi3 - hi2 * R[6]) / h32, (R[4] * hi3 - hi2 * R[7]) / h32, (pAinC[1] * hi3 - hi2 * pAinC[2]) / h32};
          const float z0 = (float)(hi1 / hi3), z1 = (float)(hi2 / hi3);
          const double r0 = (double)(un - z0), r1 = (double)(vn - z1);
          double* o = buf + lane * TRI_PITCH;
          o[0] = H[0] * H[0] + H[3] * H[3];
          o[1] = H[0] * H[1] + H[3] * H[4];
          o[2] = H[0] * H[2] + H[3] * H[5];
          o[3] = H[1] * H[1] + H[4] * H[4];
          o[4] = H[1] * H[2] + H[4] * H[5];
          o[5] = H[2] * H[2] + H[5] * H[5];
#pragma unroll
          for (int i = 0; i < 3; ++i) o[6 + i] = H[i] * r0 + H[3 + i] * r1;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        double t9[9];
        ordered_sum<9>(buf, m, t9);
        Hess[0] = t9[0];
        Hess[1] = Hess[3] = t9[1];
        Hess[2] = Hess[6] = t9[2];
        Hess[4] = t9[3];
        Hess[5] = Hess[7] = t9[4];
        Hess[8] = t9[5];
        grad[0] = t9[6];
        grad[1] = t9[7];
        grad[2] = t9[8];
      }
      double Hl[9], dx[3];
#pragma unroll
      for (int i = 0; i < 9; ++i) Hl[i] = Hess[i];
#pragma unroll
      for (int i = 0; i < 3; ++i) Hl[4 * i] *= (1.0 + lam);
      if (!tri_solve3(Hl, grad, dx)) break;
      const double cost = cost_of(alpha + dx[0], beta + dx[1], rho + dx[2]);
      if (cost <= cost_old && (cost_old - cost) / cost_old < p.min_dcost) {
        alpha += dx[0];
        beta += dx[1];
        rho += dx[2];
        eps = 0;
        break;
      }
      if (cost <= cost_old) {
        recompute = true;
        cost_old = cost;
        alpha += dx[0];
        beta += dx[1];
        rho += dx[2];
        runs++;
        lam = lam / p.lam_mult;
        eps = sqrt(dx[0] * dx[0] + dx[1] * dx[1] + dx[2] * dx[2]);
      } else {
        recompute = false;
        lam = lam * p.lam_mult;
      }
    }
    pA[0] = alpha / rho;
    pA[1] = beta / rho;
    pA[2] = 1.0 / rho;
    // largest baseline orthogonal to the bearing of the feature (max is order independent)
    const double np_ = sqrt(pA[0] * pA[0] + pA[1] * pA[1] + pA[2] * pA[2]);
    double bl = 0.0;
    if (act) {
      const double along = (pCinA[0] * pA[0] + pCinA[1] * pA[1] + pCinA[2] * pA[2]) / np_;
      const double n2 = pCinA[0] * pCinA[0] + pCinA[1] * pCinA[1] + pCinA[2] * pCinA[2] - along * along;
      bl = n2 > 0.0 ? sqrt(n2) : 0.0;
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) bl = fmax(bl, shfl_xor_f64(bl, s));
    if (pA[2] < p.min_dist || pA[2] > p.max_dist || (np_ / bl) > p.max_baseline || isnan(np_)) good = false;
  }
  if (lane == 0) {
    p.ok[f] = good ? 1 : 0;
#pragma unroll
    for (int a = 0; a < 3; ++a)
      p.p_FinG[3 * f + a] = good ? (R_GtoA[a] * pA[0] + R_GtoA[3 + a] * pA[1] + R_GtoA[6 + a] * pA[2] + p_AinG[a]) : 0.0;
  }
}

}  // namespace ovp

extern "C" hipError_t ovp_launch_triangulate(const ovp::TriParams* p, hipStream_t stream) {
  if (p->n_feats <= 0) return hipSuccess;
  hipLaunchKernelGGL(ovp::k_triangulate, dim3((p->n_feats + ovp::TRI_WAVES - 1) / ovp::TRI_WAVES), dim3(ovp::TRI_WAVES * 64), 0,
                     stream, *p);
  return hipGetLastError();
}
