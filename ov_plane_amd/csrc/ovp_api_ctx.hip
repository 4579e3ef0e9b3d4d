// C-ABI shim, part 1 (see ovp_ctx.h): context, covariance residency and bookkeeping, pose tables, feature batch, diagnostics.
// Host-side orchestration only: every arithmetic step runs in the gfx950 kernels of the k_*.hip files.
#include "ovp_ctx.h"

// ------------------------------------------------------------------------------------------------
// chi-square 0.95 quantile (replaces boost::math::quantile(chi_squared(k), 0.95),
// update/UpdaterMSCKF.cpp:59-62,749-750): regularised incomplete gamma + safeguarded Newton.
// ------------------------------------------------------------------------------------------------
static double gammap_reg(double a, double x) {
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

extern "C" double ovp_chi2_quantile_095(int dof) {
  if (dof < 1) return 0.0;
  const double p = 0.95, a = 0.5 * (double)dof;
  const double z = 1.6448536269514722;
  const double t = 1.0 - 2.0 / (9.0 * dof) + z * sqrt(2.0 / (9.0 * dof));
  double x = 0.5 * dof * t * t * t;
  if (x <= 0) x = 0.5;
  double lo = 0.0, hi = 1e300;
  for (int it = 0; it < 200; ++it) {
    const double f = gammap_reg(a, x) - p;
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

extern "C" const char* ovp_version(void) { return "ovplane_hip 0.6 (gfx950)"; }

extern "C" const char* ovp_error_string(int code) {
  switch (code) {
    case 0: return "ok";
    case OVP_E_ARG: return "bad argument";
    case OVP_E_CAPACITY: return "capacity exceeded";
    case OVP_E_NOTSPD: return "matrix not positive definite";
    case OVP_E_NEGDIAG: return "negative covariance diagonal";
    case OVP_E_NODEVICE: return "no usable HIP device";
    case OVP_E_STATE: return "call order violated";
    case OVP_E_TIMEOUT: return "device-side hand-over timed out (workgroups of the plane solve not co-resident)";
    case OVP_E_RCCL: return "RCCL not loadable or a collective call failed";
    case OVP_E_PEER: return "sharded update: the build of another rank failed (errors are collective)";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown";
  }
}


extern "C" int ovp_ctx_create(int device, int n_state_max, int n_clones_max, int n_feats_max, void* stream,
                              ovp_ctx** out) {
  if (!out || n_state_max < 1 || n_clones_max < 1 || n_clones_max > OVP_MAX_CLONES || n_feats_max < 1) return OVP_E_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= device) return OVP_E_NODEVICE;
  HIPCHK(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return OVP_E_NODEVICE;  // gfx950-only build, no fallback
  ovp_ctx* c = new ovp_ctx();
  c->device = device;
  // Two streams: the main one carries K1/K2/K3, the side stream the measurement-independent chol(P).  (Pinning the side
  // stream to one CU with hipExtStreamCreateWithCUMask was tried: the driver keeps CU masks symmetric across shader
  // engines, so removing one CU from the main stream removes 32 and K1 drops below one block per feature.)
  if (stream) {
    c->stream = (hipStream_t)stream;
  } else {
    HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->own_stream = true;
  }
  HIPCHK(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
  HIPCHK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
  for (int i = 0; i < 6; ++i) HIPCHK(hipEventCreate(&c->ev_t[i]));
  HIPCHK(hipEventCreate(&c->ev_k0));
  HIPCHK(hipEventCreate(&c->ev_k1));
  c->n_max = n_state_max;
  c->c_max = n_clones_max;
  c->f_max = n_feats_max;
  c->ld = round_up(n_state_max, 16);
  c->ldg = round_up(n_state_max + 4, 16);  // state columns | residual | 3 out-of-state plane columns
  if (c->ldg > OVP_LDG_CAP) return OVP_E_CAPACITY;  // the feature kernels stage 3 projector rows in LDS  // K1 stages the projector rows in its 64x65/2 LDS triangle
  const size_t nn = (size_t)(c->n_max + 1) * c->ld;
  HIPCHK(dalloc(&c->P, nn));
  HIPCHK(dalloc(&c->P_tmp, nn));
  HIPCHK(dalloc(&c->Ab, nn + 8));  // (+ the peer-error word the sharded update sums along with the pair)
  HIPCHK(dalloc(&c->L, nn));
  HIPCHK(dalloc(&c->W1, nn));
  HIPCHK(dalloc(&c->T, nn));
  HIPCHK(dalloc(&c->Lt, nn));
  HIPCHK(dalloc(&c->Y, nn));
  // results of an update live in ONE block [flags 4 x i32 | dx n_max | chi2 f_max | accept f_max] so that
  // ovp_msckf_fetch_results is a single device-to-host copy (four small copies cost ~5 us each)
  c->res_bytes = 16 + sizeof(double) * ((size_t)c->n_max + n_feats_max) + (size_t)n_feats_max;
  HIPCHK(hipMalloc((void**)&c->res_block, c->res_bytes));
  HIPCHK(hipMemset(c->res_block, 0, c->res_bytes));
  c->flags = (int*)c->res_block;
  c->dx = (double*)((char*)c->res_block + 16);
  c->chi2 = c->dx + c->n_max;
  c->accept = (unsigned char*)(c->chi2 + n_feats_max);
  {
    // pose tables: [R | R_fej | p | p_fej | cal(32) | clone_id | colmap], fixed offsets (capacities), uploaded as one block
    auto al = [](size_t v) { return (v + 63) & ~(size_t)63; };
    size_t o = 0;
    c->so_R = o;
    o = al(o + sizeof(double) * 9 * n_clones_max);
    c->so_Rf = o;
    o = al(o + sizeof(double) * 9 * n_clones_max);
    c->so_p = o;
    o = al(o + sizeof(double) * 3 * n_clones_max);
    c->so_pf = o;
    o = al(o + sizeof(double) * 3 * n_clones_max);
    c->so_cal = o;
    o = al(o + sizeof(double) * 32);
    c->so_id = o;
    o = al(o + sizeof(int) * n_clones_max);
    c->so_cm = o;
    o = al(o + sizeof(ovp::ColMap) * c->n_max);
    c->state_bytes = o;
    HIPCHK(hipMalloc(&c->state_block, o));
    HIPCHK(hipMemset(c->state_block, 0, o));
    HIPCHK(hipHostMalloc(&c->h_state_stage, o, hipHostMallocDefault));
    char* b = (char*)c->state_block;
    c->clone_R = (double*)(b + c->so_R);
    c->clone_R_fej = (double*)(b + c->so_Rf);
    c->clone_p = (double*)(b + c->so_p);
    c->clone_p_fej = (double*)(b + c->so_pf);
    c->cal = (double*)(b + c->so_cal);
    c->clone_id = (int*)(b + c->so_id);
    c->colmap = (ovp::ColMap*)(b + c->so_cm);
    HIPCHK(hipEventCreateWithFlags(&c->ev_state, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->ev_batch, hipEventDisableTiming));
  }
  HIPCHK(dalloc(&c->chi2_table, (size_t)OVP_CHI2_TABLE + 1));
  {
    // feature batch: [p_FinG | n_meas | clone_idx | uv], compact per upload (p_FinG always first: ovp_triangulate writes it)
    const size_t F = (size_t)n_feats_max, M = OVP_MAX_MEAS;
    c->batch_cap = sizeof(double) * 3 * F + sizeof(int) * F + sizeof(int) * F * M + sizeof(float) * 2 * F * M + 256;
    HIPCHK(hipMalloc(&c->batch_block, c->batch_cap));
    HIPCHK(hipHostMalloc(&c->h_batch_stage, c->batch_cap, hipHostMallocDefault));
    c->p_FinG = (double*)c->batch_block;
  }
  HIPCHK(dalloc(&c->G, (size_t)3 * n_feats_max * c->ldg));
  HIPCHK(dalloc(&c->Bscr, (size_t)n_feats_max * OVP_BSCR));
  HIPCHK(dalloc(&c->rec, (size_t)n_clones_max * n_feats_max * 2 * 21));
  // reduction geometry: fixed per context so the summation order (hence the result bits) is reproducible
  c->rows_per_chunk = 128;  // 32 rows = 8 MFMA steps per wave of k_gram_pair
  c->n_chunks = (2 * n_feats_max + c->rows_per_chunk - 1) / c->rows_per_chunk;
  HIPCHK(dalloc(&c->gramS, (size_t)n_clones_max * c->n_chunks * OVP_GRAM_ELEMS));
  HIPCHK(dalloc(&c->gramR, (size_t)n_clones_max * OVP_GRAM_ELEMS));
  HIPCHK(dalloc(&c->Dinv, (size_t)(c->ld / 16 + 1) * 256));
  {
    const size_t ntm = (size_t)c->ld / 16 + 1;
    HIPCHK(dalloc(&c->Ltp, ntm * (ntm + 1) / 2 * 256));  // tile-packed factor for k_fwdsub
  }
  c->n_split = (3 * n_feats_max + 63) / 64;  // split-K partials of the dense Gram product (k_gram_pair: ~256 blocks)
  if (c->n_split < 1) c->n_split = 1;
  if (c->n_split > 64) c->n_split = 64;
  {
    const int nt = c->ldg / 16;
    HIPCHK(dalloc(&c->part, (size_t)c->n_split * (nt * (nt + 1) / 2) * 256));
  }
  HIPCHK(dalloc(&c->idbuf, (size_t)4 * c->n_max + 64));
  c->small_cap = (size_t)4 * c->n_max * 64 + (size_t)c->n_max * c->n_max;
  HIPCHK(dalloc(&c->smallbuf, c->small_cap));
  HIPCHK(hipMalloc((void**)&c->ticket, 16));
  HIPCHK(hipMemset(c->ticket, 0, 16));
  HIPCHK(hipHostMalloc((void**)&c->h_res_block, c->res_bytes + 64, hipHostMallocMapped));  // pinned mirror of res_block
  memset(c->h_res_block, 0, c->res_bytes + 64);
  HIPCHK(hipHostGetDevicePointer(&c->h_res_block_dev, c->h_res_block, 0));
  HIPCHK(hipHostMalloc((void**)&c->h_slot, sizeof(int) * (size_t)(n_feats_max + 16), hipHostMallocMapped));
  HIPCHK(hipHostGetDevicePointer((void**)&c->d_slot, c->h_slot, 0));
  c->h_seq = (volatile unsigned*)((char*)c->h_res_block + ((c->res_bytes + 15) & ~(size_t)15));
  c->h_flags = (int*)c->h_res_block;
  c->h_dx = (double*)((char*)c->h_res_block + 16);
  c->h_chi2 = c->h_dx + c->n_max;
  c->h_accept = (unsigned char*)(c->h_chi2 + n_feats_max);
  // chi2 table
  {
    std::vector<double> tab(OVP_CHI2_TABLE + 1, 0.0);
    for (int k = 1; k <= OVP_CHI2_TABLE; ++k) tab[k] = ovp_chi2_quantile_095(k);
    HIPCHK(hipMemcpy(c->chi2_table, tab.data(), sizeof(double) * tab.size(), hipMemcpyHostToDevice));
  }
  memset(&c->fp, 0, sizeof(c->fp));
  *out = c;
  return 0;
}

extern "C" int ovp_ctx_destroy(ovp_ctx* c) {
  if (!c) return OVP_E_ARG;
  hipSetDevice(c->device);
  hipStreamSynchronize(c->stream);
  hipStreamSynchronize(c->stream2);
  void* dev[] = {c->P, c->P_tmp, c->Ab, c->L, c->W1, c->T, c->Lt, c->Y, c->res_block, c->ticket, c->uvn, c->tri_ok, c->state_block, c->batch_block,
                 c->chi2_table, c->G, c->Bscr, c->rec, c->gramS, c->gramR, c->Dinv, c->Ltp, c->part, c->idbuf, c->smallbuf, c->Hd, c->Acc,
                 c->bcc, c->resd, c->pl_slam_i, c->pl_slam_d, c->sub_ids, c->sub_buf, c->pl_Tbuf, c->pl_crow, c->pl_dxlast,
                 c->pl_cur, c->pl_perm, c->pl_range_done, c->pl_used, c->pl_dstage, c->pl_xbuf, c->pl_xy, c->pl_xflag, c->pl_Asum,
                 c->pl_U, c->pl_sub_tab, c->Lkeep, c->slam_res, c->slam_hscr, c->dinit_buf, c->boost, c->boost_vec};
  for (void* p : dev)
    if (p) hipFree(p);
  if (c->h_res_block) hipHostFree(c->h_res_block);
  if (c->h_slot) hipHostFree(c->h_slot);
  if (c->h_state_stage) hipHostFree(c->h_state_stage);
  if (c->h_batch_stage) hipHostFree(c->h_batch_stage);
  if (c->ev_state) hipEventDestroy(c->ev_state);
  if (c->ev_batch) hipEventDestroy(c->ev_batch);
  if (c->pl_hstage) hipHostFree(c->pl_hstage);
  if (c->pl_hres) hipHostFree(c->pl_hres);
  if (c->pl_sub_htab) hipHostFree(c->pl_sub_htab);
  if (c->ev_subtab) hipEventDestroy(c->ev_subtab);
  for (hipEvent_t e : c->pl_ev) hipEventDestroy(e);
  for (hipEvent_t e : c->pl_ev_loop) hipEventDestroy(e);
  if (c->io_h) hipHostFree(c->io_h);
  if (c->io_d) hipFree(c->io_d);
  hipEventDestroy(c->ev_fork);
  hipEventDestroy(c->ev_join);
  for (int i = 0; i < 6; ++i) hipEventDestroy(c->ev_t[i]);
  hipEventDestroy(c->ev_k0);
  hipEventDestroy(c->ev_k1);
  hipStreamDestroy(c->stream2);
  if (c->own_stream) hipStreamDestroy(c->stream);
  delete c;
  return 0;
}

extern "C" int ovp_sync(ovp_ctx* c) {
  if (!c) return OVP_E_ARG;
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

extern "C" int ovp_ctx_stream(ovp_ctx* c, void** stream) {
  if (!c || !stream) return OVP_E_ARG;
  *stream = (void*)c->stream;
  return 0;
}

extern "C" int ovp_cov_size(ovp_ctx* c) { return c ? c->n : OVP_E_ARG; }

// ---- covariance residency ----------------------------------------------------------------------
extern "C" int ovp_io_arena(ovp_ctx* c, size_t bytes, void** host, void** dev);

// (upload / download / marginal / propagate go through the pinned arena: one contiguous copy each way.  A 2-D copy from
//  pageable memory cost 90 us of host time at N = 130, a pageable copy per small array 8 us each.)
extern "C" int ovp_cov_upload(ovp_ctx* c, const double* P_host, int n, int ld) {
  drop_kept_factor(c);  // (writes the covariance: a kept factor no longer belongs to it)
  if (!c || !P_host || n < 1 || ld < n) return OVP_E_ARG;
  if (n > c->n_max) return OVP_E_CAPACITY;
  void *ah = nullptr, *ad = nullptr;
  const size_t bytes = sizeof(double) * (size_t)n * c->ld;
  {
    const int rca = ovp_io_arena(c, bytes, &ah, &ad);
    if (rca) return rca;
  }
  for (int i = 0; i < n; ++i) {
    memcpy((double*)ah + (size_t)i * c->ld, P_host + (size_t)i * ld, sizeof(double) * n);
    if (c->ld > n) memset((double*)ah + (size_t)i * c->ld + n, 0, sizeof(double) * (c->ld - n));
  }
  HIPCHK(hipMemcpyAsync(c->P, ah, bytes, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  c->n = n;
  c->have_cov = true;
  return 0;
}
extern "C" int ovp_cov_set_device(ovp_ctx* c, const double* P_dev, int n, int ld) {
  drop_kept_factor(c);  // (writes the covariance: a kept factor no longer belongs to it)
  if (!c || !P_dev || n < 1 || ld < n) return OVP_E_ARG;
  if (n > c->n_max) return OVP_E_CAPACITY;
  HIPCHK(hipMemcpy2DAsync(c->P, sizeof(double) * c->ld, P_dev, sizeof(double) * ld, sizeof(double) * n, n,
                          hipMemcpyDeviceToDevice, c->stream));
  c->n = n;
  c->have_cov = true;
  return 0;
}
extern "C" int ovp_cov_download(ovp_ctx* c, double* P_host, int n, int ld) {
  if (!c || !P_host || n != c->n || ld < n) return OVP_E_ARG;
  void *ah = nullptr, *ad = nullptr;
  const size_t bytes = sizeof(double) * (size_t)n * c->ld;
  {
    const int rca = ovp_io_arena(c, bytes, &ah, &ad);
    if (rca) return rca;
  }
  HIPCHK(hipMemcpyAsync(ah, c->P, bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  for (int i = 0; i < n; ++i) memcpy(P_host + (size_t)i * ld, (const double*)ah + (size_t)i * c->ld, sizeof(double) * n);
  return 0;
}
extern "C" int ovp_cov_marginal(ovp_ctx* c, const int* ids, const int* sizes, int n_vars, double* out_host) {
  if (!c || !ids || !sizes || !out_host || n_vars < 1) return OVP_E_ARG;
  std::vector<int> cols;
  for (int i = 0; i < n_vars; ++i)
    for (int k = 0; k < sizes[i]; ++k) {
      if (ids[i] + k >= c->n || ids[i] < 0) return OVP_E_ARG;
      cols.push_back(ids[i] + k);
    }
  const int m = (int)cols.size();
  if (m > c->n_max || (size_t)m * m > c->small_cap) return OVP_E_CAPACITY;
  void *ah = nullptr, *ad = nullptr;
  const size_t o_out = ((sizeof(int) * (size_t)m + 63) / 64) * 64, bytes = o_out + sizeof(double) * (size_t)m * m;
  {
    const int rca = ovp_io_arena(c, bytes, &ah, &ad);
    if (rca) return rca;
  }
  memcpy(ah, cols.data(), sizeof(int) * m);
  HIPCHK(hipMemcpyAsync(ad, ah, sizeof(int) * m, hipMemcpyHostToDevice, c->stream));
  HIPCHK(ovp_launch_gather_marginal(c->P, c->ld, (const int*)ad, m, (double*)((char*)ad + o_out), c->stream));
  HIPCHK(hipMemcpyAsync((char*)ah + o_out, (char*)ad + o_out, sizeof(double) * (size_t)m * m, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  memcpy(out_host, (char*)ah + o_out, sizeof(double) * (size_t)m * m);
  return 0;
}

// ---- state tables ------------------------------------------------------------------------------
static void quat_2_rot(const double q[4], double R[9]) {
  // JPL: R = (2 q4^2 - 1) I - 2 q4 [qv]x + 2 qv qv^T  (ext quat_ops.h)
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

extern "C" int ovp_state_upload(ovp_ctx* c, const ovp_state_tables* st) {
  if (!c || !st || !st->clone_q || !st->clone_p || !st->clone_q_fej || !st->clone_p_fej || !st->clone_id) return OVP_E_ARG;
  if (st->n_clones < 1 || st->n_clones > c->c_max || st->n_state > c->n_max) return OVP_E_CAPACITY;
  const int C = st->n_clones;
  for (int i = 0; i < C; ++i)
    if (st->clone_id[i] < 0 || st->clone_id[i] + 6 > st->n_state) return OVP_E_ARG;
  // everything goes through ONE pinned block and ONE copy, no synchronisation (seven pageable copies + a sync cost ~50 us,
  // more than the GPU time of a small update)
  HIPCHK(hipEventSynchronize(c->ev_state));  // the previous upload has left the staging block (normally long ago)
  char* h = (char*)c->h_state_stage;
  double* R = (double*)(h + c->so_R);
  double* Rf = (double*)(h + c->so_Rf);
  for (int i = 0; i < C; ++i) {
    quat_2_rot(st->clone_q + 4 * i, R + 9 * i);
    quat_2_rot(st->clone_q_fej + 4 * i, Rf + 9 * i);
  }
  memcpy(h + c->so_p, st->clone_p, sizeof(double) * 3 * C);
  memcpy(h + c->so_pf, st->clone_p_fej, sizeof(double) * 3 * C);
  memcpy(h + c->so_id, st->clone_id, sizeof(int) * C);
  {
    double* cal = (double*)(h + c->so_cal);
    quat_2_rot(st->calib_q, cal);
    memcpy(cal + 9, st->calib_p, sizeof(double) * 3);
    memcpy(cal + 12, st->intrinsics, sizeof(double) * 8);
  }
  // column map for the assembly kernel (calibration columns are enabled per update via the opts)
  ovp::ColMap* cm = (ovp::ColMap*)(h + c->so_cm);
  memset(cm, 0, sizeof(ovp::ColMap) * c->n_max);
  for (int i = 0; i < C; ++i)
    for (int k = 0; k < 6; ++k) {
      ovp::ColMap& m = cm[st->clone_id[i] + k];
      m.kind = 1;
      m.idx = i;
      m.off = k;
    }
  if (st->calib_id >= 0 && st->calib_id + 6 <= st->n_state)
    for (int k = 0; k < 6; ++k) {
      ovp::ColMap& m = cm[st->calib_id + k];
      m.kind = 2;
      m.idx = k;
    }
  if (st->intr_id >= 0 && st->intr_id + 8 <= st->n_state)
    for (int k = 0; k < 8; ++k) {
      ovp::ColMap& m = cm[st->intr_id + k];
      m.kind = 2;
      m.idx = 6 + k;
    }
  HIPCHK(hipMemcpyAsync(c->state_block, c->h_state_stage, c->state_bytes, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipEventRecord(c->ev_state, c->stream));
  ovp::FeatParams& fp = c->fp;
  fp.clone_R = c->clone_R;
  fp.clone_p = c->clone_p;
  fp.clone_R_fej = c->clone_R_fej;
  fp.clone_p_fej = c->clone_p_fej;
  fp.clone_id = c->clone_id;
  fp.n_clones = C;
  fp.cal = c->cal;
  c->calib_id = st->calib_id;
  c->intr_id = st->intr_id;
  c->h_clone_id.assign(st->clone_id, st->clone_id + C);
  c->fp.fisheye = st->cam_fisheye ? 1 : 0;
  c->have_state = true;
  return 0;
}

// One pinned host block + one device block per context for the entry points whose arguments are a handful of small host arrays
// (triangulation, plane fitting, plane refinement): the inputs are packed into the host block and cross the bus in ONE copy, the
// outputs come back in one.  The first versions issued a pageable copy per array (137 copy kernels per closed-loop frame with
// planes, a quarter of its GPU time) and, in the plane-fitting entries, a hipMalloc / hipFree pair per call.
extern "C" int ovp_io_arena(ovp_ctx* c, size_t bytes, void** host, void** dev) {
  if (!c || !host || !dev) return OVP_E_ARG;
  if (bytes > c->io_cap) {
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->io_h) hipHostFree(c->io_h);
    if (c->io_d) hipFree(c->io_d);
    c->io_h = c->io_d = nullptr;
    c->io_cap = 0;
    const size_t cap = bytes + bytes / 2 + 4096;
    HIPCHK(hipHostMalloc(&c->io_h, cap, hipHostMallocDefault));
    HIPCHK(hipMalloc(&c->io_d, cap));
    c->io_cap = cap;
  }
  *host = c->io_h;
  *dev = c->io_d;
  return 0;
}

// ---- feature batch -----------------------------------------------------------------------------
extern "C" int ovp_batch_upload(ovp_ctx* c, const ovp_feature_batch* b) {
  if (!c || !b || b->n_feats < 0 || b->max_meas < 1 || b->max_meas > OVP_MAX_MEAS) return OVP_E_ARG;
  if (b->n_feats > c->f_max) return OVP_E_CAPACITY;
  const size_t F = (size_t)b->n_feats, M = (size_t)b->max_meas;
  // compact layout [p_FinG | n_meas | clone_idx | uv] in one pinned block, one copy, no synchronisation: the caller's arrays
  // are free again on return because they were copied into the staging block
  auto al = [](size_t v) { return (v + 63) & ~(size_t)63; };
  const size_t o_p = 0, o_nm = al(o_p + sizeof(double) * 3 * F), o_ci = al(o_nm + sizeof(int) * F),
               o_uv = al(o_ci + sizeof(int) * F * M), total = al(o_uv + sizeof(float) * 2 * F * M);
  if (total > c->batch_cap) return OVP_E_CAPACITY;
  char* d = (char*)c->batch_block;
  c->p_FinG = (double*)(d + o_p);
  c->n_meas = (int*)(d + o_nm);
  c->clone_idx = (int*)(d + o_ci);
  c->uv = (float*)(d + o_uv);
  if (F) {
    HIPCHK(hipEventSynchronize(c->ev_batch));
    char* h = (char*)c->h_batch_stage;
    memcpy(h + o_p, b->p_FinG, sizeof(double) * 3 * F);
    memcpy(h + o_nm, b->n_meas, sizeof(int) * F);
    memcpy(h + o_ci, b->clone_idx, sizeof(int) * F * M);
    memcpy(h + o_uv, b->uv, sizeof(float) * 2 * F * M);
    HIPCHK(hipMemcpyAsync(c->batch_block, c->h_batch_stage, total, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipEventRecord(c->ev_batch, c->stream));
  }
  c->h_n_meas.assign(b->n_meas, b->n_meas + F);
  c->h_nmeas.assign(b->n_meas, b->n_meas + F);
  c->h_nmeas_valid = true;
  c->h_clone_idx.assign(b->clone_idx, b->clone_idx + F * M);
  c->fp.uv = c->uv;
  c->fp.clone_idx = c->clone_idx;
  c->fp.n_meas = c->n_meas;
  c->fp.p_FinG = c->p_FinG;
  c->n_feats = b->n_feats;
  c->max_meas = b->max_meas;
  c->have_batch = true;
  c->pl_used_valid = false;
  c->range_lo = c->range_hi = -1;
  return 0;
}
extern "C" int ovp_batch_set_range(ovp_ctx* c, int lo, int hi) {
  if (!c) return OVP_E_ARG;
  if (!c->have_batch) return OVP_E_STATE;
  if (lo == -1 && hi == -1) {
    c->range_lo = c->range_hi = -1;
    return 0;
  }
  if (lo < 0 || hi < 0 || hi > c->n_feats) return OVP_E_ARG;
  c->range_lo = lo;
  c->range_hi = hi < lo ? lo : hi;
  return 0;
}
extern "C" int ovp_batch_bind_device(ovp_ctx* c, const ovp_feature_batch* b) {
  if (!c || !b || b->n_feats < 0 || b->max_meas < 1 || b->max_meas > OVP_MAX_MEAS) return OVP_E_ARG;
  if (b->n_feats > c->f_max) return OVP_E_CAPACITY;
  c->h_n_meas.clear();
  c->h_nmeas_valid = false;  // read back lazily (once) if a caller asks for ovp_update_info
  c->h_clone_idx.clear();
  c->fp.uv = b->uv;
  c->fp.clone_idx = b->clone_idx;
  c->fp.n_meas = b->n_meas;
  c->fp.p_FinG = b->p_FinG;
  c->n_feats = b->n_feats;
  c->max_meas = b->max_meas;
  c->have_batch = true;
  c->pl_used_valid = false;
  c->range_lo = c->range_hi = -1;
  return 0;
}

extern "C" int ovp_host_timing(ovp_ctx* c, int reset, double* out8) {
  if (!c) return OVP_E_ARG;
  if (out8) memcpy(out8, c->host_acc, sizeof(c->host_acc));
  if (reset) memset(c->host_acc, 0, sizeof(c->host_acc));
  return 0;
}

// ---- propagation / clone / marginalise ---------------------------------------------------------
extern "C" int ovp_cov_propagate(ovp_ctx* c, int new_start, int phi_size, const int* old_ids, const int* old_sizes,
                                 int n_old, const double* Phi_host, const double* Q_host, int* neg_diag) {
  drop_kept_factor(c);  // (writes the covariance: a kept factor no longer belongs to it)
  if (!c || !old_ids || !old_sizes || !Phi_host || !Q_host || phi_size < 1 || n_old < 1) return OVP_E_ARG;
  if (!c->have_cov) return OVP_E_STATE;
  const int n = c->n;
  if (new_start < 0 || new_start + phi_size > n || phi_size > 64) return OVP_E_ARG;
  std::vector<int> oldcol;
  for (int i = 0; i < n_old; ++i)
    for (int k = 0; k < old_sizes[i]; ++k) {
      if (old_ids[i] < 0 || old_ids[i] + k >= n) return OVP_E_ARG;
      oldcol.push_back(old_ids[i] + k);
    }
  const int nold = (int)oldcol.size();
  if (nold > 4 * c->n_max) return OVP_E_CAPACITY;
  double* dCPT = c->smallbuf;
  double* dPCP = dCPT + (size_t)n * phi_size;
  if ((size_t)(dPCP + (size_t)phi_size * phi_size - c->smallbuf) > c->small_cap) return OVP_E_CAPACITY;
  HIPCHK(hipMemsetAsync(c->flags, 0, sizeof(int) * 4, c->stream));
  {
    // [Phi | Q | ids] packed into the pinned arena, one copy; the kernels read them from the device half of the arena
    void *ah = nullptr, *ad = nullptr;
    const size_t b_pq = sizeof(double) * ((size_t)phi_size * nold + (size_t)phi_size * phi_size);
    const size_t o_id = ((b_pq + 63) / 64) * 64, bytes = o_id + sizeof(int) * (size_t)nold;
    const int rca = ovp_io_arena(c, bytes, &ah, &ad);
    if (rca) return rca;
    memcpy(ah, Phi_host, sizeof(double) * phi_size * nold);
    memcpy((double*)ah + (size_t)phi_size * nold, Q_host, sizeof(double) * phi_size * phi_size);
    memcpy((char*)ah + o_id, oldcol.data(), sizeof(int) * nold);
    HIPCHK(hipMemcpyAsync(ad, ah, bytes, hipMemcpyHostToDevice, c->stream));
    const double* dPhi = (const double*)ad;
    const double* dQ = dPhi + (size_t)phi_size * nold;
    // the verdict comes back through mapped pinned memory (two words 16 bytes behind the point update's sequence word, in the same
    // 64-byte slack of the result block) and a sequence number of its own: no copy command, no stream synchronisation
    volatile unsigned* hw = c->h_seq + 4;
    unsigned* hw_dev = (unsigned*)((char*)c->h_res_block_dev + ((char*)hw - (char*)c->h_res_block));
    const unsigned seq = ++c->prop_seq;
    HIPCHK(ovp_launch_propagate_publish(c->P, c->ld, n, new_start, phi_size, (const int*)((char*)ad + o_id), nold, dPhi, dQ, dCPT, dPCP,
                                        c->flags + 1, hw_dev, seq, c->stream));
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (__atomic_load_n((const unsigned*)hw, __ATOMIC_ACQUIRE) != seq) {
      if ((++spins & 0xFFFu) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {
        HIPCHK(hipStreamSynchronize(c->stream));  // error path: surface a fault instead of spinning forever
        if (__atomic_load_n((const unsigned*)hw, __ATOMIC_ACQUIRE) != seq) return OVP_E_STATE;
        break;
      }
      __builtin_ia32_pause();
    }
    const int neg = (int)hw[1];
    if (neg_diag) *neg_diag = neg;
    return neg ? OVP_E_NEGDIAG : 0;
  }
}

extern "C" int ovp_cov_clone(ovp_ctx* c, int src_id, int size) {
  drop_kept_factor(c);  // (writes the covariance: a kept factor no longer belongs to it)
  if (!c || size < 1 || src_id < 0) return OVP_E_ARG;
  if (!c->have_cov) return OVP_E_STATE;
  if (src_id + size > c->n) return OVP_E_ARG;
  if (c->n + size > c->n_max) return OVP_E_CAPACITY;
  HIPCHK(ovp_launch_cov_clone(c->P, c->ld, c->n, src_id, size, c->clone_jitter, c->stream));
  c->n += size;
  return 0;
}

extern "C" int ovp_cov_clone_jitter(ovp_ctx* c, double relative_inflation) {
  if (!c || !(relative_inflation >= 0.0) || relative_inflation > 1e-6) return OVP_E_ARG;
  c->clone_jitter = relative_inflation;
  return 0;
}

extern "C" int ovp_cov_marginalize(ovp_ctx* c, int id, int size) {
  drop_kept_factor(c);  // (writes the covariance: a kept factor no longer belongs to it)
  if (!c || size < 1 || id < 0) return OVP_E_ARG;
  if (!c->have_cov) return OVP_E_STATE;
  if (id + size > c->n) return OVP_E_ARG;
  HIPCHK(ovp_launch_cov_marginalize(c->P, c->P_tmp, c->ld, c->n, id, size, c->stream));
  double* t = c->P;
  c->P = c->P_tmp;
  c->P_tmp = t;
  c->n -= size;
  return 0;
}

extern "C" int ovp_cov_initialize_invertible(ovp_ctx* c, const double* H_R, int k, int cols, int ld, const int* col_ids,
                                             const double* H_Linv, const double* R) {
  drop_kept_factor(c);  // (writes the covariance: a kept factor no longer belongs to it)
  if (!c || !H_R || !col_ids || !H_Linv || !R || k < 1 || k > 6 || cols < 1 || ld < k) return OVP_E_ARG;
  if (!c->have_cov) return OVP_E_STATE;
  const int n = c->n;
  if (n + k > c->n_max) return OVP_E_CAPACITY;
  if (cols > c->n_max) return OVP_E_ARG;
  for (int j = 0; j < cols; ++j)
    if (col_ids[j] < 0 || col_ids[j] >= n) return OVP_E_ARG;
  // device layout: H_R row-major [k][cols], Hinv / R row-major [k][k], M_a [n][6]
  std::vector<double> hr((size_t)k * cols), hi((size_t)k * k), rk((size_t)k * k);
  for (int i = 0; i < k; ++i)
    for (int a = 0; a < cols; ++a) hr[(size_t)i * cols + a] = H_R[(size_t)a * ld + i];
  for (int i = 0; i < k; ++i)
    for (int j = 0; j < k; ++j) {
      hi[(size_t)i * k + j] = H_Linv[(size_t)j * k + i];
      rk[(size_t)i * k + j] = R[(size_t)j * k + i];
    }
  double* dHR = c->smallbuf;
  double* dHi = dHR + (size_t)k * cols;
  double* dRk = dHi + 36;
  double* dMa = dRk + 36;
  if ((size_t)(dMa + (size_t)6 * n - c->smallbuf) > c->small_cap) return OVP_E_CAPACITY;
  HIPCHK(hipMemcpyAsync(dHR, hr.data(), sizeof(double) * hr.size(), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(dHi, hi.data(), sizeof(double) * hi.size(), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(dRk, rk.data(), sizeof(double) * rk.size(), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(c->idbuf, col_ids, sizeof(int) * cols, hipMemcpyHostToDevice, c->stream));
  HIPCHK(ovp_launch_init_invertible(c->P, c->ld, n, c->idbuf, cols, dHR, k, dMa, dHi, dRk, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));  // host vectors above must outlive the copies
  c->n = n + k;
  return 0;
}

extern "C" int ovp_cov_augment_dt(ovp_ctx* c, int pose_id, int dt_id, const double dnc_dt[6]) {
  drop_kept_factor(c);  // (writes the covariance: a kept factor no longer belongs to it)
  if (!c || !dnc_dt) return OVP_E_ARG;
  if (!c->have_cov) return OVP_E_STATE;
  if (pose_id < 0 || pose_id + 6 > c->n || dt_id < 0 || dt_id >= c->n) return OVP_E_ARG;
  HIPCHK(ovp_launch_augment_dt(c->P, c->ld, c->n, pose_id, dt_id, dnc_dt, c->stream));
  return 0;
}

// ---- diagnostics -------------------------------------------------------------------------------
// ---- StateHelper::initialize as one device sequence (csrc/k_init.hip) ---------------------------------------------------
extern "C" int ovp_cov_initialize(ovp_ctx* c, const double* Hx_init, const double* H_up, int k, int rup, int cols, const int* col_ids,
                                  const double* H_Linv, const double* R_init, const double* res_up, double r_iso,
                                  double chi2_threshold, int do_update, int* accepted, double* chi2, double* dx_host) {
  drop_kept_factor(c);  // (writes the covariance: a kept factor no longer belongs to it)
  if (!c || !Hx_init || !col_ids || !H_Linv || !R_init || k < 1 || k > 6 || cols < 1 || rup < 0) return OVP_E_ARG;
  if (rup > 0 && (!H_up || !res_up || !(r_iso > 0.0))) return OVP_E_ARG;
  if (!c->have_cov) return OVP_E_STATE;
  const int n = c->n, n2 = n + k, ld = c->ld, m = k + rup;
  if (n2 > c->n_max || cols > c->n_max) return OVP_E_CAPACITY;
  // S = H P H^T + R and the gathered rows of P H^T live in the LDS of one workgroup: the caller takes the three separate calls
  // for more rows than that holds (OVP_E_CAPACITY, nothing has been touched)
  if (rup > ovp_init_max_rows() || ovp_init_core_lds(k, rup, cols) > ovp_init_max_lds()) return OVP_E_CAPACITY;
  for (int j = 0; j < cols; ++j)
    if (col_ids[j] < 0 || col_ids[j] >= n) return OVP_E_ARG;
  hipStream_t s = c->stream;
  const bool upd = rup > 0 && do_update;
  // one pinned staging block: [H_all^T cols x m | Hinv 36 | Rk 36 | res rup] ids
  const size_t oHt = 0, oHi = oHt + (size_t)cols * m, oRk = oHi + 36, oRes = oRk + 36, oId = oRes + rup + 8;
  const size_t bytes = oId * sizeof(double) + sizeof(int) * (size_t)cols + 64;
  const size_t res_doubles = 4 + (size_t)c->n_max + 8;
  int rc = plane2_buffers(c, 0, bytes, res_doubles * sizeof(double));  // the plane loop's pinned staging and result blocks
  if (rc) return rc;
  double* h = (double*)c->pl_hstage;
  double* d = (double*)c->pl_dstage;
  for (int a = 0; a < cols; ++a) {
    double* row = h + oHt + (size_t)a * m;
    for (int i = 0; i < k; ++i) row[i] = Hx_init[(size_t)a * k + i];
    for (int i = 0; i < rup; ++i) row[k + i] = H_up[(size_t)a * rup + i];
  }
  for (int i = 0; i < 36; ++i) h[oHi + i] = h[oRk + i] = 0.0;
  for (int i = 0; i < k; ++i)
    for (int j = 0; j < k; ++j) {
      h[oHi + (size_t)i * k + j] = H_Linv[(size_t)j * k + i];
      h[oRk + (size_t)i * k + j] = R_init[(size_t)j * k + i];
    }
  for (int i = 0; i < rup; ++i) h[oRes + i] = res_up[i];
  memcpy(h + oId, col_ids, sizeof(int) * cols);
  const int* did = (const int*)(d + oId);
  // device scratch: result block [chi2 | accept | negdiag | - | dx n2], M_all [n2 x m], Linv [rup x rup], y [rup]
  double* dres = c->smallbuf;
  double* dM = dres + res_doubles;
  double* dLi = dM + (size_t)n2 * m;
  double* dy = dLi + (size_t)rup * rup;
  if ((size_t)(dy + rup + 8 - c->smallbuf) > c->small_cap) return OVP_E_CAPACITY;
  HIPCHK(hipMemcpyAsync(c->pl_dstage, c->pl_hstage, bytes, hipMemcpyHostToDevice, s));
  HIPCHK(ovp_launch_init_m(c->P, ld, n, did, cols, d + oHt, m, dM, s));
  HIPCHK(ovp_launch_init_core(c->P, ld, n, did, cols, d + oHt, k, rup, dM, d + oHi, d + oRk, d + oRes, r_iso > 0.0 ? r_iso : 1.0,
                              chi2_threshold, dLi, dy, dres, s));
  double* hres = (double*)c->pl_hres;
  if (upd) {
    // P+ = P - W W^T goes to the second covariance buffer (a tile reads entries other tiles overwrite)
    HIPCHK(ovp_launch_init_update(c->P, c->P_tmp, ld, n2, dM, m, k, rup, dLi, dy, dres, dres + 4, s));
  }
  {
    const int rf = ovp_fetch_to_hres(c, dres, sizeof(double) * (upd ? 4 + (size_t)n2 : 4), s);
    if (rf) return rf;
  }
  const bool ok = hres[1] > 0.5;
  if (accepted) *accepted = ok ? 1 : 0;
  if (chi2) *chi2 = hres[0];
  if (!ok) return 0;
  c->n = n2;
  if (upd) {
    double* t = c->P;
    c->P = c->P_tmp;
    c->P_tmp = t;
  }
  if (dx_host) {
    if (upd) memcpy(dx_host, hres + 4, sizeof(double) * n2);
    else memset(dx_host, 0, sizeof(double) * n2);
  }
  if (upd && hres[2] != 0.0) return OVP_E_NEGDIAG;
  return 0;
}

extern "C" long ovp_debug_read(ovp_ctx* c, const char* name, void* host, long max_bytes) {
  if (!c || !name || !host) return OVP_E_ARG;
  const size_t nn = (size_t)(c->n_max + 1) * c->ld * sizeof(double);
  const void* src = nullptr;
  size_t bytes = 0;
  if (!strcmp(name, "A") || !strcmp(name, "Ab")) { src = c->Ab; bytes = nn; }
  else if (!strcmp(name, "L")) { src = c->L; bytes = nn; }
  else if (!strcmp(name, "T")) { src = c->T; bytes = nn; }
  else if (!strcmp(name, "Lt")) { src = c->Lt; bytes = nn; }
  else if (!strcmp(name, "Y")) { src = c->Y; bytes = nn; }
  else if (!strcmp(name, "W1")) { src = c->W1; bytes = nn; }
  else if (!strcmp(name, "P")) { src = c->P; bytes = nn; }
  else if (!strcmp(name, "G")) { src = c->G; bytes = (size_t)3 * c->n_feats * c->ldg * sizeof(double); }
  else if (!strcmp(name, "rec")) { src = c->rec; bytes = (size_t)c->fp.n_clones * c->n_feats * 2 * 21 * sizeof(double); }
  else if (!strcmp(name, "plres")) { src = c->pl_res; bytes = (size_t)4 * c->pl_cap * sizeof(double); if (!src) return OVP_E_STATE; }
  else if (!strcmp(name, "An")) { src = c->pl_An; bytes = nn; if (!src) return OVP_E_STATE; }
  else if (!strcmp(name, "bn")) { src = c->pl_bn; bytes = (size_t)c->n_max * sizeof(double); if (!src) return OVP_E_STATE; }
  else if (!strcmp(name, "chi2")) {
    hipStreamSynchronize(c->stream);
    bytes = (size_t)c->n_feats * sizeof(double);
    if ((long)bytes > max_bytes) bytes = (size_t)max_bytes;
    memcpy(host, c->h_chi2, bytes);
    return (int)bytes;
  }
  else if (!strncmp(name, "bench_chol", 10)) {
    // diagnostics: average time of k_tilechol on the resident covariance; name = "bench_chol<skipmask>"
    ovp_dbg_tilechol_skip = atoi(name + 10);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) ovp_launch_tilechol(c->P, c->L, c->Dinv, c->Ltp, c->n, c->ld, c->flags + 3, 0, c->stream);
    hipEventRecord(e0, c->stream);
    for (int i = 0; i < 20; ++i) ovp_launch_tilechol(c->P, c->L, c->Dinv, c->Ltp, c->n, c->ld, c->flags + 3, 0, c->stream);
    hipEventRecord(e1, c->stream);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    ovp_dbg_tilechol_skip = 0;
    *(double*)host = ms / 20.0;
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return 8;
  }
  else if (!strcmp(name, "cycles_on")) {
    if (!c->dbg_cycles && hipMalloc((void**)&c->dbg_cycles, (size_t)c->f_max * 10 * sizeof(long long)) != hipSuccess) return OVP_E_STATE;
    return 0;
  }
  else if (!strcmp(name, "cycles")) { src = c->dbg_cycles; bytes = (size_t)c->n_feats * 10 * sizeof(long long); if (!src) return OVP_E_STATE; }
  else return OVP_E_ARG;
  if ((long)bytes > max_bytes) bytes = (size_t)max_bytes;
  if (hipStreamSynchronize(c->stream) != hipSuccess) return OVP_E_STATE;
  if (hipMemcpy(host, src, bytes, hipMemcpyDeviceToHost) != hipSuccess) return OVP_E_STATE;
  return (long)bytes;
}

// Diagnostics / micro-benchmark of the second-generation tile Cholesky (k_chol2.hip): factorizes the n x n host matrix A (+ I)
// bordered with the row brow, returns the dense factor of the bordered matrix ((n+1) x (n+1) row-major, or n x n without a border),
// z = L^-1 brow, y = L^-T z and the pivots; avg_ms = average duration of `reps` launches (HIP events).
static double g_dbg_chol2_floor = 0.0;
extern "C" void ovp_debug_chol2_floor(double piv_floor) { g_dbg_chol2_floor = piv_floor; }

extern "C" int ovp_debug_chol2(ovp_ctx* c, const double* A_host, int n, int lda, const double* brow_host, int add_identity,
                               double* L_host, double* z_host, double* y_host, double* piv_host, int reps, float* avg_ms) {
  if (!c || !A_host || n < 1 || lda < n) return OVP_E_ARG;
  const int nb = brow_host ? n + 1 : n;
  if (nb > ovp_chol2_max_n() + 1) return OVP_E_CAPACITY;
  double *dA = nullptr, *dL = nullptr, *dv = nullptr;
  HIPCHK(dalloc(&dA, (size_t)n * n));
  HIPCHK(dalloc(&dL, (size_t)nb * nb));
  HIPCHK(dalloc(&dv, (size_t)4 * n + 16));
  HIPCHK(hipMemcpy2D(dA, sizeof(double) * n, A_host, sizeof(double) * lda, sizeof(double) * n, n, hipMemcpyHostToDevice));
  if (brow_host) HIPCHK(hipMemcpy(dv, brow_host, sizeof(double) * n, hipMemcpyHostToDevice));
  HIPCHK(hipMemset(dL, 0, sizeof(double) * (size_t)nb * nb));
  HIPCHK(hipMemsetAsync(c->flags, 0, sizeof(int) * 4, c->stream));
  ovp::Chol2Job j;
  memset(&j, 0, sizeof(j));
  j.A = dA;
  j.n = n;
  j.ld = n;
  j.add_identity = add_identity;
  j.mode = 0;
  j.brow = brow_host ? dv : nullptr;
  j.flag = c->flags;
  j.Ldense = dL;
  j.ldo = nb;
  j.z_out = brow_host ? dv + n : nullptr;
  j.y_out = brow_host ? dv + 2 * n : nullptr;
  j.piv_out = dv + 3 * n;
  j.piv_floor = g_dbg_chol2_floor;
  HIPCHK(ovp_launch_chol2(&j, nullptr, nullptr, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (getenv("OVP_C2_STAMPS")) {
    long long* st = nullptr;
    HIPCHK(hipMalloc((void**)&st, sizeof(long long) * 16 * 32));
    HIPCHK(hipMemset(st, 0, sizeof(long long) * 16 * 32));
    ovp::Chol2Job jt = j;
    jt.Ldense = nullptr;
    if (!getenv("OVP_C2_TIME_Y")) jt.y_out = nullptr;
    jt.stamps = st;
    HIPCHK(ovp_launch_chol2(&jt, nullptr, nullptr, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    long long h[16 * 32];
    HIPCHK(hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost));
    hipFree(st);
    const int nt = (nb + 15) / 16;
    fprintf(stderr, "chol2 stamps (cycles): elimination wave 0 [wait column | read+eliminate | write] ; tile wave 0 [wait panel | reload + next column | wait buffer + publish | rest]\n");
    fprintf(stderr, " tile wave 0 prologue: issue loads %lld, patch special tiles %lld, publish column 0 %lld (elimination wave 0 starts waiting at %lld after the tile wave)\n",
            h[13] - h[16 + 13], h[14] - h[13], h[15] - h[14], h[0] - h[16 + 13]);
    for (int k = 0; k < nt; ++k) {
      const long long* e = h + k * 16;
      fprintf(stderr, " k=%2d E: %6lld %6lld %6lld | T: %6lld %6lld %6lld %6lld | E step %6lld T step %6lld | on arrival: column %+lld panel %+lld trail %+lld | E start %lld T start %lld\n", k, e[1] - e[0], e[2] - e[1],
              e[3] - e[2], e[9] - e[8], e[10] - e[9], e[11] - e[10], e[12] - e[11], e[3] - e[0], e[12] - e[8], e[4] / 1000000 - 500, (e[4] / 1000) % 1000 - 500, e[4] % 1000 - 500, e[0] - h[0], e[8] - h[0]);
    }
    if (jt.y_out) {
      fprintf(stderr, "back substitution: preparation (sub-diagonal tiles to LDS, inverses of the diagonal blocks) %lld cycles + barrier %lld, chain %lld\n",
              h[nt * 16 + 4] - h[nt * 16 + 3], h[nt * 16 + 5] - h[nt * 16 + 4], h[7] - h[nt * 16 + 5]);
      fprintf(stderr, "back substitution, wave 0 per step: [first product + loads | wait for the partial sums | sum, second product, publish]\n");
      for (int k = nt - 1; k >= 0; --k) {
        const long long* e = h + k * 16;
        fprintf(stderr, " k=%2d  %6lld %6lld %6lld | step %6lld\n", k, e[5] - e[4], k <= nt - 3 ? e[6] - e[5] : 0LL,
                e[7] - (k <= nt - 3 ? e[6] : e[5]), e[7] - e[4]);
      }
    }
  }
  if (reps > 0) {
    ovp::Chol2Job jt = j;  // timing: the factorization alone (no dense output)
    jt.Ldense = nullptr;
    if (!getenv("OVP_C2_TIME_Y")) jt.y_out = nullptr;  // OVP_C2_TIME_Y: with the back substitution
    jt.dbg = getenv("OVP_C2_DBG") ? atoi(getenv("OVP_C2_DBG")) : 0;
    HIPCHK(hipEventRecord(c->ev_t[0], c->stream));
    for (int r = 0; r < reps; ++r) HIPCHK(ovp_launch_chol2(&jt, nullptr, nullptr, c->stream));
    HIPCHK(hipEventRecord(c->ev_t[1], c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    float ms = 0.f;
    hipEventElapsedTime(&ms, c->ev_t[0], c->ev_t[1]);
    if (avg_ms) *avg_ms = ms / reps;
  }
  if (L_host) HIPCHK(hipMemcpy(L_host, dL, sizeof(double) * (size_t)nb * nb, hipMemcpyDeviceToHost));
  if (z_host && brow_host) HIPCHK(hipMemcpy(z_host, dv + n, sizeof(double) * n, hipMemcpyDeviceToHost));
  if (y_host && brow_host) HIPCHK(hipMemcpy(y_host, dv + 2 * n, sizeof(double) * n, hipMemcpyDeviceToHost));
  if (piv_host) HIPCHK(hipMemcpy(piv_host, dv + 3 * n, sizeof(double) * n, hipMemcpyDeviceToHost));
  int fl[4];
  HIPCHK(hipMemcpy(fl, c->flags, sizeof(fl), hipMemcpyDeviceToHost));
  HIPCHK(hipMemsetAsync(c->flags, 0, sizeof(int) * 4, c->stream));
  hipFree(dA);
  hipFree(dL);
  hipFree(dv);
  return (fl[0] & 2) ? OVP_E_TIMEOUT : (fl[0] ? OVP_E_NOTSPD : 0);
}

extern "C" int ovp_last_timings(ovp_ctx* c, float* ms4) {
  if (!c || !ms4) return OVP_E_ARG;
  memcpy(ms4, c->last_ms, sizeof(float) * 4);
  return 0;
}

extern "C" int ovp_plane_kernel_timer(ovp_ctx* c, int enable, int reset, float* avg_ms, int* n_launches) {
  if (!c) return OVP_E_ARG;
  if (avg_ms) *avg_ms = c->pl_klaunches ? (float)(c->pl_ktime_ms / c->pl_klaunches) : 0.f;
  if (n_launches) *n_launches = c->pl_klaunches;
  if (reset) {
    c->pl_ktime_ms = 0.0;
    c->pl_klaunches = 0;
  }
  c->pl_ktimer = enable;
  return 0;
}

extern "C" int ovp_kernel_timer(ovp_ctx* c, int enable, int reset, float* avg_ms_feat, int* n_launches) {
  if (!c) return OVP_E_ARG;
  if (avg_ms_feat) *avg_ms_feat = c->klaunches ? (float)(c->ktime_ms / c->klaunches) : 0.f;
  if (n_launches) *n_launches = c->klaunches;
  if (reset) {
    c->ktime_ms = 0.0;
    c->klaunches = 0;
  }
  c->ktimer = enable != 0;
  return 0;
}

