// C-ABI shim, part 5 (see ovp_ctx.h): SLAM landmarks (update/UpdaterSLAM.cpp:66-682) and triangulation.
#include "ovp_ctx.h"

// ---- triangulation ----------------------------------------------------------------
extern "C" void ovp_triang_defaults(ovp_triang_opts* o) {
  if (!o) return;
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

extern "C" int ovp_triangulate(ovp_ctx* c, const ovp_triang_opts* o, const float* uv_norm, double* p_FinG_out, uint8_t* ok) {
  if (!c || !o || !uv_norm || !ok) return OVP_E_ARG;
  if (!c->have_state || !c->have_batch) return OVP_E_STATE;
  const size_t F = (size_t)c->n_feats, M = (size_t)c->max_meas;
  if (F == 0) return 0;
  // arena: [uv_norm | -> p_FinG | ok]
  auto al = [](size_t v) { return (v + 63) & ~(size_t)63; };
  const size_t b_uv = sizeof(float) * F * M * 2, o_p = al(b_uv), o_ok = al(o_p + sizeof(double) * 3 * F), total = al(o_ok + F);
  void *ah = nullptr, *ad = nullptr;
  {
    const int rca = ovp_io_arena(c, total, &ah, &ad);
    if (rca) return rca;
  }
  memcpy(ah, uv_norm, b_uv);
  HIPCHK(hipMemcpyAsync(ad, ah, b_uv, hipMemcpyHostToDevice, c->stream));
  ovp::TriParams tp;
  tp.uvn = (const float*)ad;
  tp.clone_idx = c->fp.clone_idx;
  tp.n_meas = c->fp.n_meas;
  tp.n_feats = (int)F;
  tp.max_meas = (int)M;
  tp.clone_R = c->clone_R;
  tp.clone_p = c->clone_p;
  tp.cal = c->cal;
  tp.refine_features = o->refine_features;
  tp.triangulate_1d = o->triangulate_1d;
  tp.max_runs = o->max_runs;
  tp.init_lamda = o->init_lamda;
  tp.max_lamda = o->max_lamda;
  tp.min_dx = o->min_dx;
  tp.min_dcost = o->min_dcost;
  tp.lam_mult = o->lam_mult;
  tp.min_dist = o->min_dist;
  tp.max_dist = o->max_dist;
  tp.max_baseline = o->max_baseline;
  tp.max_cond_number = o->max_cond_number;
  tp.p_FinG = c->p_FinG;  // the library's own buffer even when the batch was bound to caller memory
  tp.ok = (unsigned char*)ad + o_ok;
  HIPCHK(ovp_launch_triangulate(&tp, c->stream));
  c->fp.p_FinG = c->p_FinG;
  // results into the pinned block (the positions stay in the batch's buffer on the device as linearisation points)
  HIPCHK(hipMemcpyAsync((char*)ah + o_p, c->p_FinG, sizeof(double) * 3 * F, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync((char*)ah + o_ok, (char*)ad + o_ok, F, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (p_FinG_out) memcpy(p_FinG_out, (char*)ah + o_p, sizeof(double) * 3 * F);
  memcpy(ok, (char*)ah + o_ok, F);
  return 0;
}

// ---- UpdaterSLAM::update on the device (update/UpdaterSLAM.cpp:424-673; csrc/k_slam.hip) ------------------------------------
// Rows and gate of every landmark in ONE launch against the resident covariance (no download of P, no host gate), the accepted rows
// stacked on the device, StateHelper::EKFUpdate on that stack (S-form up to 80 rows, information form above), one synchronisation.
extern "C" int ovp_slam_update(ovp_ctx* c, const ovp_update_opts* o, const ovp_slam_batch* b, double* dx_host, uint8_t* status_host,
                               double* chi2_host, ovp_update_info* info) {
  drop_kept_factor(c);  // (writes the covariance: a kept factor no longer belongs to it)
  if (!c || !o || !b || b->n_landmarks < 0) return OVP_E_ARG;
  if (!c->have_state || !c->have_cov) return OVP_E_STATE;
  const int L = b->n_landmarks, n = c->n, M = b->max_meas;
  if (info) memset(info, 0, sizeof(*info));
  if (dx_host) memset(dx_host, 0, sizeof(double) * n);
  if (L == 0) return 0;
  if (M < 1 || M > OVP_MAX_MEAS || !b->n_meas || !b->landmark_id) return OVP_E_ARG;
  const bool any_pre = b->pre_rows != nullptr;
  if (any_pre && (!b->pre_cols || !b->pre_H || !b->pre_ids)) return OVP_E_ARG;
  const unsigned calmask = (o->do_calib_camera_pose ? 0x3Fu : 0u) | (o->do_calib_camera_intrinsics ? (0xFFu << 6) : 0u);
  int calcol[14];
  for (int k = 0; k < 14; ++k) {
    calcol[k] = (k < 6) ? c->calib_id + k : c->intr_id + (k - 6);
    if (!((calmask >> k) & 1)) calcol[k] = 0;
    else if (calcol[k] < 0 || calcol[k] >= n) return OVP_E_ARG;
  }
  const int C = (int)c->h_clone_id.size();
  // ---- host: the call's column list (first-seen order, as Hx_order_big of :634-646), row offsets, kernel geometry
  std::vector<int> gpos(n, -1), gids, row0(L), pre_off(L, 0), pre_ids_off(L, 0);
  auto touch = [&](int col) {
    if (gpos[col] < 0) {
      gpos[col] = (int)gids.size();
      gids.push_back(col);
    }
  };
  int m_total = 0, rows_max = 1, cols_max = 1;
  size_t preH = 0, preI = 0;
  bool any_built = false;
  for (int l = 0; l < L; ++l) {
    row0[l] = m_total;
    int rows, cols;
    if (any_pre && b->pre_rows[l] > 0) {
      rows = b->pre_rows[l];
      cols = b->pre_cols[l];
      if (cols < 1 || cols > n) return OVP_E_ARG;
      pre_off[l] = (int)preH;
      pre_ids_off[l] = (int)preI;
      for (int k = 0; k < cols; ++k) {
        const int id = b->pre_ids[preI + k];
        if (id < 0 || id >= n) return OVP_E_ARG;
        touch(id);
      }
      preH += (size_t)rows * cols + rows;
      preI += cols;
    } else {
      if (!b->uv || !b->clone_idx || !b->p_FinG || !b->p_FinG_fej) return OVP_E_ARG;
      const int m = b->n_meas[l];
      if (m < 0 || m > M) return OVP_E_ARG;
      const bool plane = b->plane_state_id && b->plane_state_id[l] >= 0;
      if (plane && (!b->cp || !b->cp_fej || b->plane_state_id[l] + 3 > n)) return OVP_E_ARG;
      if (b->landmark_id[l] < 0 || b->landmark_id[l] + 3 > n) return OVP_E_ARG;
      rows = plane ? 3 * m : 2 * m;
      cols = 6 * m + __builtin_popcount(calmask) + 3 + (plane ? 3 : 0);
      for (int a = 0; a < m; ++a) {
        const int ci = b->clone_idx[(size_t)l * M + a];
        if (ci < 0 || ci >= C) return OVP_E_ARG;
        for (int k = 0; k < 6; ++k) touch(c->h_clone_id[ci] + k);
      }
      if (m > 0) {
        for (int k = 0; k < 14; ++k)
          if ((calmask >> k) & 1) touch(calcol[k]);
        for (int k = 0; k < 3; ++k) touch(b->landmark_id[l] + k);
        if (plane)
          for (int k = 0; k < 3; ++k) touch(b->plane_state_id[l] + k);
      }
      any_built = any_built || m > 0;
    }
    m_total += rows;
    rows_max = std::max(rows_max, rows);
    cols_max = std::max(cols_max, cols);
  }
  if (m_total < 1) {  // nothing to update with (:661-663)
    if (status_host) memset(status_host, 0, L);
    if (chi2_host) memset(chi2_host, 0, sizeof(double) * L);
    return 0;
  }
  const int gcols = (int)gids.size();
  if (ovp_slam_gate_lds(rows_max, cols_max, 0) > 150 * 1024) return OVP_E_CAPACITY;
  const int h_in_lds = ovp_slam_gate_lds(rows_max, cols_max, 1) <= 150 * 1024 ? 1 : 0;
  hipStream_t s = c->stream;
  // ---- one pinned staging block -> one copy
  auto al = [](size_t v) { return (v + 63) & ~(size_t)63; };
  size_t off = 0;
  auto take = [&](size_t bytes) {
    const size_t o0 = off;
    off = al(off + bytes);
    return o0;
  };
  const size_t o_p = take(sizeof(double) * 3 * L), o_pf = take(sizeof(double) * 3 * L), o_cp = take(sizeof(double) * 3 * L),
               o_cpf = take(sizeof(double) * 3 * L), o_preH = take(sizeof(double) * (preH + 1)), o_uv = take(sizeof(float) * 2 * (size_t)L * M),
               o_ci = take(sizeof(int) * (size_t)L * M), o_nm = take(sizeof(int) * L), o_lm = take(sizeof(int) * L),
               o_ps = take(sizeof(int) * L), o_r0 = take(sizeof(int) * L), o_gp = take(sizeof(int) * n),
               o_gi = take(sizeof(int) * gcols), o_pr = take(sizeof(int) * L), o_pc = take(sizeof(int) * L),
               o_po = take(sizeof(int) * L), o_pio = take(sizeof(int) * L), o_pid = take(sizeof(int) * (preI + 1));
  const size_t stage_bytes = off;
  const size_t res_doubles = 4 + (size_t)c->n_max + 8;
  const size_t lres_bytes = al(sizeof(double) * L) + al((size_t)L);
  int rc = plane2_buffers(c, 0, stage_bytes, res_doubles * sizeof(double) + lres_bytes + 64);
  if (rc) return rc;
  char* h = (char*)c->pl_hstage;
  char* d = (char*)c->pl_dstage;
  memset(h, 0, stage_bytes);
  if (any_built) {
    memcpy(h + o_p, b->p_FinG, sizeof(double) * 3 * L);
    memcpy(h + o_pf, b->p_FinG_fej, sizeof(double) * 3 * L);
    memcpy(h + o_uv, b->uv, sizeof(float) * 2 * (size_t)L * M);
    memcpy(h + o_ci, b->clone_idx, sizeof(int) * (size_t)L * M);
  }
  if (b->cp) memcpy(h + o_cp, b->cp, sizeof(double) * 3 * L);
  if (b->cp_fej) memcpy(h + o_cpf, b->cp_fej, sizeof(double) * 3 * L);
  memcpy(h + o_nm, b->n_meas, sizeof(int) * L);
  memcpy(h + o_lm, b->landmark_id, sizeof(int) * L);
  for (int l = 0; l < L; ++l) ((int*)(h + o_ps))[l] = b->plane_state_id ? b->plane_state_id[l] : -1;
  memcpy(h + o_r0, row0.data(), sizeof(int) * L);
  memcpy(h + o_gp, gpos.data(), sizeof(int) * n);
  memcpy(h + o_gi, gids.data(), sizeof(int) * gcols);
  if (any_pre) {
    memcpy(h + o_pr, b->pre_rows, sizeof(int) * L);
    memcpy(h + o_pc, b->pre_cols, sizeof(int) * L);
    memcpy(h + o_po, pre_off.data(), sizeof(int) * L);
    memcpy(h + o_pio, pre_ids_off.data(), sizeof(int) * L);
    memcpy(h + o_preH, b->pre_H, sizeof(double) * preH);
    memcpy(h + o_pid, b->pre_ids, sizeof(int) * preI);
  }
  // ---- device buffers: the stacked system (Hd = H^T [gcols][m_total], resd), per-landmark results, block scratch
  const size_t need = (size_t)gcols * m_total;
  if (need > c->Hd_cap) {
    if (c->Hd) hipFree(c->Hd);
    c->Hd = nullptr;
    c->Hd_cap = 0;
    HIPCHK(dalloc(&c->Hd, need + 64));
    c->Hd_cap = need + 64;
  }
  if ((size_t)m_total > c->res_cap) {
    if (c->resd) hipFree(c->resd);
    c->resd = nullptr;
    c->res_cap = 0;
    HIPCHK(dalloc(&c->resd, (size_t)m_total + 64));
    c->res_cap = (size_t)m_total + 64;
  }
  if (lres_bytes > c->slam_res_cap) {
    if (c->slam_res) hipFree(c->slam_res);
    c->slam_res = nullptr;
    c->slam_res_cap = 0;
    HIPCHK(hipMalloc(&c->slam_res, lres_bytes + 4096));
    c->slam_res_cap = lres_bytes + 4096;
  }
  if (!h_in_lds) {
    const size_t hs = (size_t)L * rows_max * cols_max;
    if (hs > c->slam_hscr_cap) {
      if (c->slam_hscr) hipFree(c->slam_hscr);
      c->slam_hscr = nullptr;
      c->slam_hscr_cap = 0;
      HIPCHK(dalloc(&c->slam_hscr, hs + 64));
      c->slam_hscr_cap = hs + 64;
    }
  }
  HIPCHK(hipMemcpyAsync(d, h, stage_bytes, hipMemcpyHostToDevice, s));
  ovp::SlamParams sp;
  memset(&sp, 0, sizeof(sp));
  sp.fp = c->fp;
  sp.fp.uv = (const float*)(d + o_uv);
  sp.fp.clone_idx = (const int*)(d + o_ci);
  sp.fp.n_meas = (const int*)(d + o_nm);
  sp.fp.p_FinG = (const double*)(d + o_p);
  sp.fp.n_feats = L;
  sp.fp.max_meas = M;
  sp.fp.do_fej = o->do_fej;
  sp.fp.calmask = calmask;
  for (int k = 0; k < 14; ++k) sp.fp.calcol[k] = calcol[k];
  sp.fp.white_px = 1.0 / o->sigma_px;
  sp.fp.chi2_mult = o->chi2_multiplier;
  sp.fp.chi2_table = c->chi2_table;
  sp.fp.P = c->P;
  sp.fp.n = n;
  sp.fp.ldp = c->ld;
  sp.p_fej = (const double*)(d + o_pf);
  sp.lm_id = (const int*)(d + o_lm);
  sp.plane_sid = (const int*)(d + o_ps);
  sp.cp = (const double*)(d + o_cp);
  sp.cp_fej = (const double*)(d + o_cpf);
  sp.white_c = 1.0 / o->sigma_constraint;
  if (any_pre) {
    sp.pre_rows = (const int*)(d + o_pr);
    sp.pre_cols = (const int*)(d + o_pc);
    sp.pre_off = (const int*)(d + o_po);
    sp.pre_ids_off = (const int*)(d + o_pio);
    sp.pre_H = (const double*)(d + o_preH);
    sp.pre_ids = (const int*)(d + o_pid);
  }
  sp.row0 = (const int*)(d + o_r0);
  sp.gpos = (const int*)(d + o_gp);
  sp.Ht = c->Hd;
  sp.m_total = m_total;
  sp.gcols = gcols;
  sp.res_out = c->resd;
  sp.Hscr = c->slam_hscr;
  sp.rows_max = rows_max;
  sp.cols_max = cols_max;
  sp.h_in_lds = h_in_lds;
  const char* form_env = getenv("OVP_EKF_INFO_FORM");
  const bool info_form_only = form_env && form_env[0] == '1';
  // S-form (k_init.hip) up to 80 stacked rows: scratch [res 4 | dx n_max | 8 | chi2 L | status L] M_all | Linv | y in smallbuf, so
  // that everything the host wants comes back in ONE copy
  const bool sform = !info_form_only && m_total <= ovp_init_max_rows() && ovp_init_core_lds(0, m_total, gcols) <= ovp_init_max_lds();
  double* dres = c->smallbuf;
  double* dM = (double*)((char*)(dres + res_doubles) + lres_bytes);
  double* dLi = dM + (size_t)n * m_total;
  double* dy = dLi + (size_t)m_total * m_total;
  const bool sform_fits = (size_t)(dy + m_total + 8 - c->smallbuf) <= c->small_cap;
  if (sform && sform_fits) {
    sp.chi2 = dres + res_doubles;
    sp.status = (unsigned char*)(dres + res_doubles) + al(sizeof(double) * L);
    sp.Mall = dM;
  } else {
    sp.chi2 = (double*)c->slam_res;
    sp.status = (unsigned char*)c->slam_res + al(sizeof(double) * L);
  }
  HIPCHK(ovp_launch_slam_gate(&sp, L, ovp_slam_gate_lds(rows_max, cols_max, h_in_lds), s));
  const int* dgid = (const int*)(d + o_gi);
  char* hres = (char*)c->pl_hres;
  double* hres_d = (double*)hres;
  char* hl = hres + res_doubles * sizeof(double);  // [chi2 L | status L]
  auto finish_landmarks = [&]() {
    if (chi2_host) memcpy(chi2_host, hl, sizeof(double) * L);
    if (status_host) memcpy(status_host, hl + al(sizeof(double) * L), L);
    if (info) {
      info->n_cols = gcols;
      for (int l = 0; l < L; ++l) {
        const unsigned char st = ((unsigned char*)(hl + al(sizeof(double) * L)))[l];
        if (!st) continue;
        info->n_accepted++;
        const int rows_l = (l + 1 < L ? row0[l + 1] : m_total) - row0[l];
        const bool pre = any_pre && b->pre_rows[l] > 0;
        info->n_rows += (st == 2 && !pre) ? 2 * b->n_meas[l] : rows_l;
      }
    }
  };
  if (sform && sform_fits) {
    const int rows = m_total;
    HIPCHK(ovp_launch_init_core(c->P, c->ld, n, dgid, gcols, c->Hd, 0, rows, dM, c->resd /* unused: k = 0 */, c->resd, c->resd, 1.0, 1e300,
                                dLi, dy, dres, s));
    HIPCHK(ovp_launch_init_update(c->P, c->P_tmp, c->ld, n, dM, rows, 0, rows, dLi, dy, dres, dres + 4, s));
    {
      const int rf = ovp_fetch_to_hres(c, dres, res_doubles * sizeof(double) + lres_bytes, s);
      if (rf) return rf;
    }
    finish_landmarks();
    if (info) {
      info->not_spd = hres_d[1] > 0.5 ? 0 : 1;
      info->neg_diag = hres_d[2] != 0.0;
    }
    if (!(hres_d[1] > 0.5)) return OVP_E_NOTSPD;  // S = H P H^T + I lost definiteness: P is not a covariance; nothing was written
    double* t = c->P;
    c->P = c->P_tmp;
    c->P_tmp = t;
    if (dx_host) memcpy(dx_host, hres_d + 4, sizeof(double) * n);
    return hres_d[2] != 0.0 ? OVP_E_NEGDIAG : 0;
  }
  // information form: A = H^T H, b = H^T r on the call's columns, scattered to the state
  if (!c->Acc) HIPCHK(dalloc(&c->Acc, (size_t)c->n_max * c->n_max));
  if (!c->bcc) HIPCHK(dalloc(&c->bcc, (size_t)c->n_max));
  HIPCHK(hipMemsetAsync(c->flags, 0, sizeof(int) * 4, s));
  HIPCHK(ovp_launch_gemm(0, 1, gcols, gcols, m_total, c->Hd, m_total, c->Hd, m_total, c->Acc, gcols, 0, s));
  HIPCHK(ovp_launch_gemm(0, 0, gcols, 1, m_total, c->Hd, m_total, c->resd, 1, c->bcc, 1, 0, s));
  HIPCHK(hipMemsetAsync(c->Ab, 0, sizeof(double) * (size_t)(n + 1) * c->ld, s));
  HIPCHK(ovp_launch_scatter_gram(c->Acc, c->bcc, gcols, dgid, c->Ab, c->ld, n, s));
  {
    std::vector<int> ids(gids);
    std::sort(ids.begin(), ids.end());
    int rs = set_substate(c, ids);
    if (rs) return rs;
  }
  rc = ekf_from_gram(c, false);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(c->h_dx, c->dx, sizeof(double) * n, hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(c->h_flags, c->flags, sizeof(int) * 4, hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(hl, c->slam_res, lres_bytes, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  finish_landmarks();
  if (c->h_flags[0]) {  // positive semi-definite prior: S-form instead of the factor of P
    int rs = ekf_sform(c);
    if (rs) return rs;
  }
  if (dx_host) memcpy(dx_host, c->h_dx, sizeof(double) * n);
  if (info) {
    info->not_spd = c->h_flags[0];
    info->neg_diag = c->h_flags[1];
  }
  if (c->h_flags[0]) return OVP_E_NOTSPD;
  if (c->h_flags[1]) return OVP_E_NEGDIAG;
  return 0;
}

// ---- UpdaterSLAM::delayed_init, candidate loop on the device (update/UpdaterSLAM.cpp:204-364; csrc/k_dinit.hip) -----------------
extern "C" int ovp_slam_delayed_init(ovp_ctx* c, const ovp_update_opts* o, const ovp_feature_batch* b, uint8_t* ok_host,
                                     double* chi2_host, int* new_id, double* delta_init, double* dx_host, int dx_stride) {
  drop_kept_factor(c);  // (writes the covariance: a kept factor no longer belongs to it)
  if (!c || !o || !b || b->n_feats < 0) return OVP_E_ARG;
  if (!c->have_state || !c->have_cov) return OVP_E_STATE;
  const int L = b->n_feats, M = b->max_meas, n0 = c->n, ld = c->ld;
  if (L == 0) return 0;
  if (M < 2 || M > OVP_MAX_MEAS || !b->uv || !b->clone_idx || !b->n_meas || !b->p_FinG) return OVP_E_ARG;
  if (dx_host && dx_stride < n0 + 3 * L) return OVP_E_ARG;
  if (n0 + 3 * L > c->n_max) return OVP_E_CAPACITY;
  const unsigned calmask = (o->do_calib_camera_pose ? 0x3Fu : 0u) | (o->do_calib_camera_intrinsics ? (0xFFu << 6) : 0u);
  const int ncal = __builtin_popcount(calmask);
  int calcol[14];
  for (int k = 0; k < 14; ++k) {
    calcol[k] = (k < 6) ? c->calib_id + k : c->intr_id + (k - 6);
    if (!((calmask >> k) & 1)) calcol[k] = 0;
    else if (calcol[k] < 0 || calcol[k] >= n0) return OVP_E_ARG;
  }
  const int C = (int)c->h_clone_id.size();
  int cols_max = 1, rows_max = 4;
  for (int l = 0; l < L; ++l) {
    const int m = b->n_meas[l];
    if (m < 2 || m > M) return OVP_E_ARG;  // (update/UpdaterSLAM.cpp:112-118: the caller drops shorter tracks)
    for (int a = 0; a < m; ++a) {
      const int ci = b->clone_idx[(size_t)l * M + a];
      if (ci < 0 || ci >= C) return OVP_E_ARG;
    }
    const int cols = 6 * m + ncal, rup = 2 * m - 3;
    // outside the one-workgroup S-form (k_init.hip): the caller takes StateHelper::initialize candidate by candidate; nothing touched
    if (rup > ovp_init_max_rows() || ovp_init_core_lds(3, rup, cols) > ovp_init_max_lds() ||
        ovp_dinit_rows_lds(m, ncal) > OVP_DINIT_DYN_LDS) return OVP_E_CAPACITY;
    cols_max = std::max(cols_max, cols);
    rows_max = std::max(rows_max, 2 * m);
  }
  hipStream_t s = c->stream;
  auto al = [](size_t v) { return (v + 63) & ~(size_t)63; };
  // staging: the candidates as a feature batch + their column lists
  size_t off = 0;
  auto take = [&](size_t bytes) {
    const size_t o0 = off;
    off = al(off + bytes);
    return o0;
  };
  const size_t o_p = take(sizeof(double) * 3 * L), o_uv = take(sizeof(float) * 2 * (size_t)L * M), o_ci = take(sizeof(int) * (size_t)L * M),
               o_nm = take(sizeof(int) * L), o_id = take(sizeof(int) * (size_t)L * cols_max);
  const size_t stage_bytes = off;
  const size_t res_doubles = 4 + (size_t)c->n_max + 8;
  int rc = plane2_buffers(c, 0, stage_bytes, sizeof(double) * res_doubles * L + 64);
  if (rc) return rc;
  char* h = (char*)c->pl_hstage;
  char* d = (char*)c->pl_dstage;
  memcpy(h + o_p, b->p_FinG, sizeof(double) * 3 * L);
  memcpy(h + o_uv, b->uv, sizeof(float) * 2 * (size_t)L * M);
  memcpy(h + o_ci, b->clone_idx, sizeof(int) * (size_t)L * M);
  memcpy(h + o_nm, b->n_meas, sizeof(int) * L);
  for (int l = 0; l < L; ++l) {
    int* ids = (int*)(h + o_id) + (size_t)l * cols_max;
    const int m = b->n_meas[l];
    for (int a = 0; a < m; ++a)
      for (int k = 0; k < 6; ++k) ids[6 * a + k] = c->h_clone_id[b->clone_idx[(size_t)l * M + a]] + k;
    int q = 6 * m;
    for (int k = 0; k < 14; ++k)
      if ((calmask >> k) & 1) ids[q++] = calcol[k];
  }
  // device scratch: [result blocks L x res_doubles | Ht | Mall | Linv | y | Hinv 9 | Rk 9 | resid]
  const size_t n_end = (size_t)n0 + 3 * L;
  const size_t need = res_doubles * L + (size_t)cols_max * rows_max + n_end * rows_max + (size_t)rows_max * rows_max + rows_max + 32 +
                      rows_max + 64;
  if (need > c->dinit_cap) {
    if (c->dinit_buf) hipFree(c->dinit_buf);
    c->dinit_buf = nullptr;
    c->dinit_cap = 0;
    HIPCHK(dalloc(&c->dinit_buf, need + 1024));
    c->dinit_cap = need + 1024;
  }
  double* dres0 = c->dinit_buf;
  double* dHt = dres0 + res_doubles * L;
  double* dM = dHt + (size_t)cols_max * rows_max;
  double* dLi = dM + n_end * rows_max;
  double* dy = dLi + (size_t)rows_max * rows_max;
  double* dHinv = dy + rows_max + 8;
  double* dRk = dHinv + 12;
  double* dresid = dRk + 12;
  HIPCHK(hipMemcpyAsync(d, h, stage_bytes, hipMemcpyHostToDevice, s));
  ovp::DinitParams dp;
  memset(&dp, 0, sizeof(dp));
  dp.fp = c->fp;
  dp.fp.uv = (const float*)(d + o_uv);
  dp.fp.clone_idx = (const int*)(d + o_ci);
  dp.fp.n_meas = (const int*)(d + o_nm);
  dp.fp.p_FinG = (const double*)(d + o_p);
  dp.fp.n_feats = L;
  dp.fp.max_meas = M;
  dp.fp.do_fej = o->do_fej;
  dp.fp.calmask = calmask;
  for (int k = 0; k < 14; ++k) dp.fp.calcol[k] = calcol[k];
  dp.fp.white_px = 1.0 / o->sigma_px;
  dp.fp.ldp = ld;
  dp.n_max = c->n_max;
  dp.P = c->P;
  dp.clone_R = c->clone_R;
  dp.clone_p = c->clone_p;
  dp.cal = c->cal;
  dp.Ht = dHt;
  dp.Hinv = dHinv;
  dp.Rk = dRk;
  dp.resid = dresid;
  for (int l = 0; l < L; ++l) {
    const int m = b->n_meas[l], cols = 6 * m + ncal, rows = 2 * m, rup = rows - 3, n = n0 + 3 * l;
    dp.cand = l;
    dp.m_obs = m;
    dp.n = n;
    dp.prev_res = l ? dres0 + res_doubles * (l - 1) : nullptr;
    dp.ids = (const int*)(d + o_id) + (size_t)l * cols_max;
    memcpy(dp.idv, (const int*)(h + o_id) + (size_t)l * cols_max, sizeof(int) * cols);
    dp.res = dres0 + res_doubles * l;
    HIPCHK(ovp_launch_dinit_rows(&dp, ovp_dinit_rows_lds(m, ncal), s));
    HIPCHK(ovp_launch_init_m(c->P, ld, n, dp.ids, cols, dHt, rows, dM, s));  // M = P[:, ids] H_all^T on many workgroups
    // chi2 of the update rows with dof = all rows (StateHelper.cpp:471), initialize_invertible, update in place
    const double thr = o->chi2_multiplier * ovp_chi2_quantile_095(rows);
    HIPCHK(ovp_launch_init_core(c->P, ld, n, dp.ids, cols, dHt, 3, rup, dM, dHinv, dRk, dresid, 1.0, thr, dLi, dy, dp.res, s));
    HIPCHK(ovp_launch_init_update(c->P, c->P, ld, n + 3, dM, rows, 3, rup, dLi, dy, dp.res, dp.res + 4, s));
  }
  dp.cand = -1;
  dp.n = (int)n_end;
  dp.prev_res = dres0 + res_doubles * (L - 1);
  HIPCHK(ovp_launch_dinit_rows(&dp, 64, s));
  double* hres = (double*)c->pl_hres;
  {
    const int rf = ovp_fetch_to_hres(c, dres0, sizeof(double) * res_doubles * L, s);
    if (rf) return rf;
  }
  c->n = (int)n_end;
  // final layout: the inert blocks of the rejected candidates go (last first), the accepted ones move up
  std::vector<int> final_id(L, -1);
  int n_acc = 0, negdiag = 0;
  for (int l = 0; l < L; ++l) {
    const double* r = hres + res_doubles * l;
    if (r[1] > 0.5) final_id[l] = n0 + 3 * n_acc++;
    if (r[1] > 0.5 && r[2] != 0.0) negdiag = 1;
  }
  for (int l = L - 1; l >= 0; --l)
    if (final_id[l] < 0) {
      rc = ovp_cov_marginalize(c, n0 + 3 * l, 3);
      if (rc) return rc;
    }
  for (int l = 0; l < L; ++l) {
    const double* r = hres + res_doubles * l;
    const bool ok = r[1] > 0.5;
    if (ok_host) ok_host[l] = ok ? 1 : 0;
    if (chi2_host) chi2_host[l] = r[0];
    if (new_id) new_id[l] = final_id[l];
    if (delta_init)
      for (int k = 0; k < 3; ++k) delta_init[3 * l + k] = ok ? r[4 + c->n_max + k] : 0.0;
    if (dx_host) {
      double* dx = dx_host + (size_t)l * dx_stride;
      memset(dx, 0, sizeof(double) * dx_stride);
      if (ok) {
        memcpy(dx, r + 4, sizeof(double) * n0);
        for (int g = 0; g <= l; ++g)  // the landmarks that were state variables at that point, at their final ids
          if (final_id[g] >= 0)
            for (int k = 0; k < 3; ++k) dx[final_id[g] + k] = r[4 + n0 + 3 * g + k];
      }
    }
  }
  return negdiag ? OVP_E_NEGDIAG : 0;
}

