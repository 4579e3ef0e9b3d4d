// C-ABI shim, part 4 (see ovp_ctx.h): the per-plane loop of UpdaterMSCKF::update (update/UpdaterMSCKF.cpp:411-649,
// update/UpdaterPlane.cpp:296-552) and UpdaterPlane::init_vio_plane.
#include "ovp_ctx.h"

// device sequence shared by the plane update and the plane initialisation: feature kernel, Gram reduction, reduction to the
// state columns, range-energy factorisation, information-form update with the factor Mf (P = Mf Mf^T), gate.
// Leaves: V in c->Y, dx in c->dx, [chi2, ok, n_deg, pr] in c->pl_res + 4*pl, the extended Gram in c->pl_E.
static int plane_job_device(ovp_ctx* c, const ovp_update_opts* o, const ovp::FeatParams& fp, int pl, int start, int nf,
                            int in_state, int sid, double white_c, const double* Mf, int factor_dense, double thr,
                            int rows_live, int rows_u, int n_involved, int force = -1) {
  const int n = c->n, ld = c->ld, ldg = c->ldg;
  hipStream_t s = c->stream;
  ovp::PlaneParams pp;
  pp.feat_list = c->pl_featlist + start;
  pp.n_local = nf;
  pp.plane = pl;
  pp.in_state = in_state;
  pp.plane_sid = sid;
  pp.white_c = white_c;
  pp.cp = c->pl_cp;
  pp.cp_fej = c->pl_cp_fej;
  pp.cst = c->pl_cst;
  ovp::FeatParams fpl = fp;
  fpl.n = n;
  fpl.P = c->P;
  HIPCHK(ovp_launch_plane_feat(&fpl, &pp, nf, s));
  const int chunks = (2 * nf + c->rows_per_chunk - 1) / c->rows_per_chunk;
  HIPCHK(ovp_launch_struct_gram(c->rec, fp.n_clones, nf, c->rows_per_chunk, chunks, c->gramS, s));
  HIPCHK(ovp_launch_reduce_gram(c->gramS, fp.n_clones, chunks, c->gramR, s));
  int nsplit = (3 * nf + 511) / 512;
  if (nsplit < 1) nsplit = 1;
  if (nsplit > c->n_split) nsplit = c->n_split;
  HIPCHK(ovp_launch_syrk(c->G, 3 * nf, ldg, n + 4, nsplit, c->part, s));
  HIPCHK(ovp_launch_reduce_cst(c->pl_cst, nf, c->pl_cstsum, s));
  HIPCHK(ovp_launch_assemble_ext(c->gramR, fp.n_clones, c->part, nsplit, c->colmap, n, sid, c->pl_cstsum, c->pl_E, ldg, s));
  if (!in_state && c->pl_n_slam > 0)  // landmarks lying on this plane: one constraint row each (UpdaterMSCKF.cpp:545-552)
    HIPCHK(ovp_launch_plane_slam_rows(c->pl_E, ldg, n, pl + 1, c->pl_n_slam, c->pl_slam_i, c->pl_slam_i + c->pl_slam_cap,
                                      c->pl_slam_d, c->pl_slam_d + 3 * (size_t)c->pl_slam_cap, c->pl_cp + 3 * pl,
                                      c->pl_cp_fej + 3 * pl, white_c, fp.do_fej, c->pl_cstsum, s));
  HIPCHK(ovp_launch_plane_reduce_to_state(c->pl_E, ldg, n, in_state, c->Ab, ld, c->pl_cstsum + 9, c->pl_scal, s));
  // range part of the residual (regularised, diagonally normalised): its own Cholesky, independent of the update's -
  // side stream, joined before the gate (the two write different words of pl_scal)
  hipStream_t s2 = c->stream2;
  HIPCHK(hipEventRecord(c->ev_fork, s));
  HIPCHK(hipStreamWaitEvent(s2, c->ev_fork, 0));
  HIPCHK(ovp_launch_normalize_reg(c->Ab, ld, n, 1e-10, c->pl_An, c->pl_bn, s2));
  HIPCHK(ovp_launch_tilechol(c->pl_An, c->pl_Lr, c->pl_Dinv2, nullptr, n, ld, c->flags + 2, 0, s2));
  HIPCHK(ovp_launch_range_energy(c->pl_Lr, c->pl_Dinv2, c->pl_bn, n, ld, 1e-8, c->pl_scal, s2));
  HIPCHK(hipEventRecord(c->ev_join, s2));
  // EKF update in information form with the chained factor
  HIPCHK(ovp_launch_gemm4(0, 0, n, n, n, c->Ab, ld, Mf, ld, c->W1, ld, 0, 0, s));
  HIPCHK(ovp_launch_gemm4(1, 0, n, n, n, Mf, ld, c->W1, ld, c->T, ld, 1, 1, s));
  HIPCHK(chol_of_T(c, c->T, n, ld, 0, nullptr, s));  // (second-generation factorization where it fits, like every other chol(T))
  HIPCHK(ovp_launch_fwdsub(c->Ltp, c->Dinv, Mf, c->Y, n, ld, factor_dense, s));
  HIPCHK(ovp_launch_dx_from_factor(c->Y, n, ld, c->Ab + (size_t)n * ld, c->dx, c->pl_scal, s));
  HIPCHK(hipStreamWaitEvent(s, c->ev_join, 0));
  HIPCHK(ovp_launch_plane_gate(c->pl_scal, c->flags, thr, rows_live, rows_u, n_involved, force, c->pl_res + 4 * pl, s));
  return 0;
}

static int plane_buffers(ovp_ctx* c, int NP) {
  const int ld = c->ld;
  if (NP > c->pl_cap || !c->pl_E) {
    void* olds[] = {c->pl_sid, c->pl_cp, c->pl_cp_fej, c->pl_res, c->pl_dx};
    for (void* p : olds)
      if (p) hipFree(p);
    const int cap = NP + 8;
    HIPCHK(dalloc(&c->pl_sid, (size_t)cap));
    HIPCHK(dalloc(&c->pl_cp, (size_t)3 * cap));
    HIPCHK(dalloc(&c->pl_cp_fej, (size_t)3 * cap));
    HIPCHK(dalloc(&c->pl_res, (size_t)4 * cap));
    HIPCHK(dalloc(&c->pl_dx, (size_t)c->n_max * cap));
    c->pl_cap = cap;
    if (!c->pl_E) {
      const size_t ne = (size_t)(c->n_max + 4) * c->ldg;
      HIPCHK(dalloc(&c->pl_featlist, (size_t)c->f_max));
      HIPCHK(dalloc(&c->pl_cst, (size_t)c->f_max * 10));
      HIPCHK(dalloc(&c->pl_cstsum, 16));
      HIPCHK(dalloc(&c->pl_E, ne));
      HIPCHK(dalloc(&c->pl_An, (size_t)(c->n_max + 1) * ld));
      HIPCHK(dalloc(&c->pl_bn, (size_t)c->n_max));
      HIPCHK(dalloc(&c->pl_Lr, (size_t)(c->n_max + 1) * ld));
      HIPCHK(dalloc(&c->pl_Dinv2, (size_t)(ld / 16 + 1) * 256));
      HIPCHK(dalloc(&c->pl_scal, 8));
    }
  }
  return 0;
}

// ---- UpdaterMSCKF::update, per-plane loop (second generation) ------------------------------------------------------------
// See k_plane2.hip for the algebra.  Everything of a call is enqueued without a host synchronisation: the per-call tables go
// through one pinned staging block, the results come back through one pinned block read after a single stream sync.
struct PlaneJobH { int pl, start, nf, rows_total, rows_live, rows_u, n_involved, in_state, sid, n_inv_cols, ns_pl; double thr; };

extern "C" int ovp_msckf_plane_update(ovp_ctx* c, const ovp_update_opts* o, const ovp_plane_batch* pb, double* dx_planes,
                                      uint8_t* plane_ok, double* plane_chi2, int* plane_dof, uint8_t* feat_used);

// The plane loop's results go to the host as the point update's do (ovp_api_point.hip: k_publish_results): ONE kernel behind the
// loop writes [chi2, decision, ... per plane | dx per plane | consumed features | flags] into mapped pinned memory and then a
// sequence number the host spins on - instead of four copy commands (a blit kernel of ~4 us each on the stream) and a stream
// synchronisation (interrupt + wake-up).  It also clears the device flags for the next call (was a fill command).
__global__ __launch_bounds__(1024) void k_publish_plane_results(const double* __restrict__ res, int n_res, const double* __restrict__ dx,
                                                               int n_dx, const unsigned char* __restrict__ used, int n_used,
                                                               int* __restrict__ flags, double* __restrict__ dst_res,
                                                               double* __restrict__ dst_dx, unsigned char* __restrict__ dst_used,
                                                               int* __restrict__ dst_flags, volatile unsigned* seq_host, unsigned seq) {
  const int t = threadIdx.x;
  for (int i = t; i < n_res; i += 1024) dst_res[i] = res[i];
  for (int i = t; i < n_dx; i += 1024) dst_dx[i] = dx[i];
  {
    // (8 bytes at a time: `used` and its mirror are 8-byte aligned, the tail byte by byte)
    const int w = n_used >> 3;
    const unsigned long long* us = reinterpret_cast<const unsigned long long*>(used);
    unsigned long long* ud = reinterpret_cast<unsigned long long*>(dst_used);
    for (int i = t; i < w; i += 1024) ud[i] = us[i];
    for (int i = 8 * w + t; i < n_used; i += 1024) dst_used[i] = used[i];
  }
  if (t < 4) dst_flags[t] = flags[t];
  __threadfence_system();
  __syncthreads();
  if (t == 0) *seq_host = seq;
  if (t < 4) flags[t] = 0;
}

// A device block -> the start of the pinned result block (c->pl_hres), by a kernel + sequence word instead of a copy command and a
// stream synchronisation (the entry points with a handful of results: SLAM update, delayed initialisation, initialize, dense update).
// Falls back to the copy when the block has no room for the sequence word.
__global__ __launch_bounds__(1024) void k_fetch_block(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, int bytes,
                                                     volatile unsigned* seq_host, unsigned seq) {
  const int w = bytes >> 3;
  const unsigned long long* s8 = reinterpret_cast<const unsigned long long*>(src);
  unsigned long long* d8 = reinterpret_cast<unsigned long long*>(dst);
  for (int i = threadIdx.x; i < w; i += 1024) d8[i] = s8[i];
  for (int i = 8 * w + threadIdx.x; i < bytes; i += 1024) dst[i] = src[i];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) *seq_host = seq;
}
int ovp_fetch_to_hres(ovp_ctx* c, const void* dsrc, size_t bytes, hipStream_t s) {
  if (!c->pl_hres_dev || bytes + 64 > c->pl_hres_cap || (((size_t)dsrc) & 7) || bytes > (size_t)0x7fffffff) {
    HIPCHK(hipMemcpyAsync(c->pl_hres, dsrc, bytes, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return 0;
  }
  const size_t o_seq = (c->pl_hres_cap - 64) & ~(size_t)63;
  volatile unsigned* hseq = (volatile unsigned*)((char*)c->pl_hres + o_seq);
  const unsigned seq = ++c->pl_pub_seq;
  hipLaunchKernelGGL(k_fetch_block, dim3(1), dim3(1024), 0, s, (const unsigned char*)dsrc, (unsigned char*)c->pl_hres_dev, (int)bytes,
                     (volatile unsigned*)((char*)c->pl_hres_dev + o_seq), seq);
  HIPCHK(hipGetLastError());
  const auto t0 = std::chrono::steady_clock::now();
  unsigned spins = 0;
  while (__atomic_load_n((const unsigned*)hseq, __ATOMIC_ACQUIRE) != seq) {
    if ((++spins & 0xFFFu) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {
      HIPCHK(hipStreamSynchronize(s));  // error path: surface a fault instead of spinning forever
      if (__atomic_load_n((const unsigned*)hseq, __ATOMIC_ACQUIRE) != seq) return OVP_E_STATE;
      break;
    }
    __builtin_ia32_pause();
  }
  return 0;
}

static int ensure_pl_used(ovp_ctx* c) {
  if (!c->pl_used) HIPCHK(hipMalloc((void**)&c->pl_used, (size_t)c->f_max + 16));
  return 0;
}

int plane2_buffers(ovp_ctx* c, int NP, size_t stage_bytes, size_t res_bytes) {
  const int ld = c->ld;
  if (!c->pl_Tbuf) {
    const size_t nn = (size_t)(c->n_max + 1) * ld;
    HIPCHK(dalloc(&c->pl_Tbuf, 2 * nn));
    HIPCHK(dalloc(&c->pl_crow, (size_t)c->n_max + 16));
    HIPCHK(dalloc(&c->pl_dxlast, (size_t)c->n_max + 16));
    HIPCHK(hipMalloc((void**)&c->pl_cur, 16));
    HIPCHK(hipMalloc((void**)&c->pl_range_done, 16));
    HIPCHK(hipMemset(c->pl_range_done, 0, 16));
    HIPCHK(dalloc(&c->pl_xbuf, (size_t)9 * 18 * 256));
    HIPCHK(dalloc(&c->pl_xy, (size_t)c->n_max + 32));
    HIPCHK(hipMalloc((void**)&c->pl_xflag, sizeof(unsigned) * 64));
    HIPCHK(hipMemset(c->pl_xflag, 0, sizeof(unsigned) * 64));
  }
  if (NP > c->pl2_cap) {
    if (c->pl_perm) hipFree(c->pl_perm);
    c->pl2_cap = NP + 8;
    HIPCHK(hipMalloc((void**)&c->pl_perm, sizeof(int) * (size_t)c->pl2_cap * c->n_max));
  }
  if (stage_bytes > c->pl_stage_cap) {
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->pl_hstage) hipHostFree(c->pl_hstage);
    if (c->pl_dstage) hipFree(c->pl_dstage);
    c->pl_stage_cap = stage_bytes + 4096;
    HIPCHK(hipHostMalloc(&c->pl_hstage, c->pl_stage_cap, hipHostMallocDefault));
    HIPCHK(hipMalloc(&c->pl_dstage, c->pl_stage_cap));
  }
  if (res_bytes > c->pl_hres_cap) {
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->pl_hres) hipHostFree(c->pl_hres);
    c->pl_hres_cap = res_bytes + 4096;
    HIPCHK(hipHostMalloc(&c->pl_hres, c->pl_hres_cap, hipHostMallocMapped));
    memset(c->pl_hres, 0, c->pl_hres_cap);
    HIPCHK(hipHostGetDevicePointer(&c->pl_hres_dev, c->pl_hres, 0));
  }
  return 0;
}

// ---- a selection of state columns as a state of its own (plane loop in its own order, plane initialisation on the marginal) ----
// Device block [ids | inverse | clone ids | column map] of the selection `ids` (pos = its inverse, -1 = not selected), staged in the
// context's pinned block and sent on stream s; the kernels then address the selection through these tables instead of the state's.
struct SubTables {
  const int* d_ids = nullptr;
  const int* d_inv = nullptr;
  int* d_clone_id = nullptr;
  ovp::ColMap* d_colmap = nullptr;
  int calib_sub = -1, intr_sub = -1;
  std::vector<int> clone_sub;
};
static int sub_tables_upload(ovp_ctx* c, const ovp_update_opts* o, const std::vector<int>& ids, const std::vector<int>& pos,
                             SubTables* t, hipStream_t s) {
  const int n = c->n, C = c->fp.n_clones, ns = (int)ids.size();
  const size_t o_ids = 0, o_inv = sizeof(int) * (size_t)(c->n_max + 16), o_tab = 2 * o_inv;
  const size_t tab_bytes = sizeof(int) * (size_t)(c->c_max + 16) + sizeof(ovp::ColMap) * (size_t)c->n_max;
  const size_t blk_bytes = o_tab + tab_bytes;
  if (!c->pl_sub_tab) {
    HIPCHK(hipMalloc(&c->pl_sub_tab, blk_bytes));
    HIPCHK(hipHostMalloc(&c->pl_sub_htab, blk_bytes, hipHostMallocDefault));
  }
  if (c->ev_subtab) HIPCHK(hipEventSynchronize(c->ev_subtab));  // the pinned block fed the copy of the previous call (long done)
  else HIPCHK(hipEventCreateWithFlags(&c->ev_subtab, hipEventDisableTiming));
  char* hb = (char*)c->pl_sub_htab;
  memset(hb, 0, blk_bytes);
  int* h_ids = (int*)(hb + o_ids);
  int* h_inv = (int*)(hb + o_inv);
  int* t_clone = (int*)(hb + o_tab);
  ovp::ColMap* t_cm = (ovp::ColMap*)(hb + o_tab + sizeof(int) * (size_t)(c->c_max + 16));
  memcpy(h_ids, ids.data(), sizeof(int) * (size_t)ns);
  for (int col = 0; col < n; ++col) h_inv[col] = pos[col] >= 0 ? pos[col] : 0;
  t->clone_sub.assign((size_t)C, 0);
  for (int i = 0; i < C; ++i) {
    t->clone_sub[i] = t_clone[i] = pos[c->h_clone_id[i]];
    for (int k = 0; k < 6; ++k) {
      ovp::ColMap& m = t_cm[t->clone_sub[i] + k];
      m.kind = 1;
      m.idx = i;
      m.off = k;
    }
  }
  t->calib_sub = (c->calib_id >= 0 && c->calib_id + 6 <= n && pos[c->calib_id] >= 0) ? pos[c->calib_id] : -1;
  t->intr_sub = (c->intr_id >= 0 && c->intr_id + 8 <= n && pos[c->intr_id] >= 0) ? pos[c->intr_id] : -1;
  if (t->calib_sub >= 0 && o->do_calib_camera_pose)
    for (int k = 0; k < 6; ++k) {
      t_cm[t->calib_sub + k].kind = 2;
      t_cm[t->calib_sub + k].idx = k;
    }
  if (t->intr_sub >= 0 && o->do_calib_camera_intrinsics)
    for (int k = 0; k < 8; ++k) {
      t_cm[t->intr_sub + k].kind = 2;
      t_cm[t->intr_sub + k].idx = 6 + k;
    }
  HIPCHK(hipMemcpyAsync(c->pl_sub_tab, hb, blk_bytes, hipMemcpyHostToDevice, s));
  HIPCHK(hipEventRecord(c->ev_subtab, s));
  t->d_ids = (const int*)((char*)c->pl_sub_tab + o_ids);
  t->d_inv = (const int*)((char*)c->pl_sub_tab + o_inv);
  t->d_clone_id = (int*)((char*)c->pl_sub_tab + o_tab);
  t->d_colmap = (ovp::ColMap*)((char*)c->pl_sub_tab + o_tab + sizeof(int) * (size_t)(c->c_max + 16));
  return 0;
}
// the context's view of the state while a selection stands in for it, and back
struct SubSaved {
  int n, calib_id, intr_id;
  double* P;
  int* clone_id;
  ovp::ColMap* colmap;
  const int* fp_clone_id;
  std::vector<int> h_clone_id;
};
static SubSaved sub_enter(ovp_ctx* c, const SubTables& t, int ns, double* Psub) {
  SubSaved sv{c->n, c->calib_id, c->intr_id, c->P, c->clone_id, c->colmap, c->fp.clone_id, c->h_clone_id};
  c->n = ns;
  c->P = Psub;
  c->calib_id = t.calib_sub;
  c->intr_id = t.intr_sub;
  c->clone_id = t.d_clone_id;
  c->fp.clone_id = c->clone_id;
  c->colmap = t.d_colmap;
  c->h_clone_id = t.clone_sub;
  return sv;
}
static void sub_leave(ovp_ctx* c, const SubSaved& sv) {
  c->n = sv.n;
  c->P = sv.P;
  c->calib_id = sv.calib_id;
  c->intr_id = sv.intr_id;
  c->clone_id = sv.clone_id;
  c->fp.clone_id = sv.fp_clone_id;
  c->colmap = sv.colmap;
  c->h_clone_id = sv.h_clone_id;
}

// ---- the plane loop in the loop's own column order (update/UpdaterMSCKF.cpp:413-649 has no size limit) --------------------------
// A plane's rows touch the clones, the calibration, its own closest point when it is a state variable and the SLAM landmarks lying
// on it (out-of-state planes).  Two things follow:
//  (1) LEADING BLOCK.  With P0 = L0 L0^T in the order [clones + calibration | the planes' own columns in processing order |
//      everything no plane of the call involves (IMU, dt, other landmarks)], A_k is zero outside the columns involved so far and L0
//      is lower triangular, so L0^T A_k L0 is zero outside that LEADING block: T_k = blockdiag(T_lead, I).  Plane k's products and
//      its factorization run on nl_k = 6 C + calibration + (own columns of the planes up to k) columns instead of n (config 3:
//      194 .. 224 of 240 - 13 to 14 tile steps of k_chol2 instead of 15, and shorter ones); only dx = L0[:, 0:nl] y and the commit
//      see all n rows.  The covariance is permuted once in front of the loop and once behind it.
//  (2) SUB-STATE.  Above the factorization's limit (n > 287) the loop runs on the marginal P0[s, s] of the involved columns s
//      (ns <= 287; same order) - same kernels, the state tables addressed through remapped column ids - and the rest of the state
//      follows from the push-through identity (k_plane_sub_accum for dx, the point path's  P -= G (A - A Pss+ A) G^T  for P).
static int plane_update_ordered(ovp_ctx* c, const ovp_update_opts* o, const ovp_plane_batch* pb, double* dx_planes,
                                uint8_t* plane_ok, double* plane_chi2, int* plane_dof, uint8_t* feat_used) {
  const double t_entry = host_now_ms();
  const int n = c->n, ld = c->ld, NP = pb->n_planes, C = c->fp.n_clones;
  const int n_slam = pb->n_slam > 0 ? pb->n_slam : 0;
  if (n_slam > 0 && (!pb->slam_plane || !pb->slam_state_id || !pb->slam_p || !pb->slam_p_fej)) return OVP_E_ARG;
  // ---- column order: first involvement ----
  std::vector<int> ids, pos((size_t)n, -1);
  ids.reserve((size_t)n);
  bool bad_id = false;
  auto place = [&](int id, int sz) {
    if (id < 0 || id + sz > n) {
      bad_id = true;
      return;
    }
    for (int k = 0; k < sz; ++k)
      if (pos[id + k] < 0) {
        pos[id + k] = (int)ids.size();
        ids.push_back(id + k);
      }
  };
  for (int i = 0; i < C; ++i) place(c->h_clone_id[i], 6);
  if (o->do_calib_camera_pose) place(c->calib_id, 6);
  if (o->do_calib_camera_intrinsics) place(c->intr_id, 8);
  for (int q = 0; q < n_slam; ++q)
    if (pb->slam_plane[q] < 1 || pb->slam_plane[q] > NP || pb->slam_state_id[q] < 0 || pb->slam_state_id[q] + 3 > n) return OVP_E_ARG;
  c->pl_nl.assign((size_t)(NP > 0 ? NP : 1), 0);
  for (int k = 0; k < NP; ++k) {
    if (pb->plane_state_id[k] >= 0) place(pb->plane_state_id[k], 3);
    else
      for (int q = 0; q < n_slam; ++q)
        if (pb->slam_plane[q] == k + 1) place(pb->slam_state_id[q], 3);
    c->pl_nl[k] = (int)ids.size();
  }
  if (bad_id) return OVP_E_ARG;
  const int n_inv = (int)ids.size();
  const bool full = n <= ovp_chol2_max_n();  // the whole state fits one factorization: the rest rides along behind the leading block
  if (!full && n_inv > ovp_chol2_max_n()) return OVP_E_CAPACITY;  // the planes of this call involve more columns than one factorization holds
  if (full)
    for (int col = 0; col < n; ++col)
      if (pos[col] < 0) {
        pos[col] = (int)ids.size();
        ids.push_back(col);
      }
  const int ns = (int)ids.size();
  hipStream_t s = c->stream;
  if (!full) {
    if (!c->pl_Asum) HIPCHK(dalloc(&c->pl_Asum, (size_t)c->n_max * ld));
    if (NP > c->pl_U_cap) {
      if (c->pl_U) hipFree(c->pl_U);
      c->pl_U_cap = NP + 8;
      HIPCHK(dalloc(&c->pl_U, (size_t)c->pl_U_cap * ld));
    }
  }
  std::vector<int> sid_sub(NP > 0 ? NP : 1, -1), slam_sub(n_slam > 0 ? n_slam : 1, 0);
  for (int k = 0; k < NP; ++k) sid_sub[k] = pb->plane_state_id[k] >= 0 ? pos[pb->plane_state_id[k]] : -1;
  for (int q = 0; q < n_slam; ++q) {
    // a landmark listed on a plane that IS in the state takes no part in the loop (UpdaterMSCKF.cpp:240-241): park it on column 0
    // (in the full order every column has a position; on a marginal the landmark's columns may be absent)
    const int p0 = (pb->slam_state_id[q] >= 0 && pb->slam_state_id[q] + 3 <= n) ? pos[pb->slam_state_id[q]] : -1;
    if (p0 < 0 && pb->plane_state_id[pb->slam_plane[q] - 1] < 0) return OVP_E_ARG;
    slam_sub[q] = p0 >= 0 ? p0 : 0;
  }
  if (c->pl_ktimer) {
    while (c->pl_ev_loop.size() < 2) {
      hipEvent_t e;
      HIPCHK(hipEventCreate(&e));
      c->pl_ev_loop.push_back(e);
    }
    HIPCHK(hipEventRecord(c->pl_ev_loop[0], s));
  }
  // ---- remapped tables: [ids | inverse | clone ids | column map] ----
  SubTables st;
  {
    const int rt = sub_tables_upload(c, o, ids, pos, &st, s);
    if (rt) return rt;
  }
  const int* d_ids = st.d_ids;
  const int* d_inv = st.d_inv;
  // full order: the columns behind the involved ones take a diagonal boost that the un-permutation behind the loop takes off
  // again (k_gather_block_boost) - an exact stochastic clone then factors at the first attempt
  c->pl_boost_active = full && n_inv < ns;
  if (c->pl_boost_active) {
    if (!c->boost_vec) HIPCHK(dalloc(&c->boost_vec, (size_t)c->n_max + 16));
    HIPCHK(ovp_launch_gather_block_boost(c->P, ld, d_ids, ns, c->P_tmp, ld, n_inv, 1e-9, c->boost_vec, s));
  } else {
    HIPCHK(ovp_launch_gather_block(c->P, ld, d_ids, ns, c->P_tmp, ld, s));
  }
  if (!full) {
    HIPCHK(hipMemsetAsync(c->pl_Asum, 0, sizeof(double) * (size_t)ns * ld, s));
    HIPCHK(hipMemsetAsync(c->pl_U, 0, sizeof(double) * (size_t)NP * ld, s));
  }
  // ---- the loop in the new order ----
  ovp_plane_batch pbs = *pb;
  pbs.plane_state_id = sid_sub.data();
  pbs.slam_state_id = slam_sub.data();
  std::vector<double> dx_sub((size_t)ns * (NP > 0 ? NP : 1), 0.0);
  SubSaved sv = sub_enter(c, st, ns, c->P_tmp);
  c->pl_sub_active = true;
  c->pl_sub_rest = !full;
  c->pl_scatter_dst = full ? sv.P : nullptr;  // full order: the loop's covariance product is un-permuted straight into the resident P
  c->pl_scatter_ids = d_inv;
  c->pl_t_entry = t_entry;
  const int rc = ovp_msckf_plane_update(c, o, &pbs, dx_sub.data(), plane_ok, plane_chi2, plane_dof, feat_used);
  c->pl_sub_active = false;
  c->pl_sub_rest = false;
  c->pl_scatter_dst = nullptr;
  double* Pss_new = c->P;  // = P_tmp: the marginal after the loop
  sub_leave(c, sv);
  if (rc) return rc;  // the resident covariance was not touched (the device tables may have been: a loop that fails after
                      // accepting planes has marked them invalid, have_state = false - INTEGRATION.md section 5)
  if (full) {
    if (dx_planes)
      for (int k = 0; k < NP; ++k)
        for (int i = 0; i < ns; ++i) dx_planes[(size_t)k * n + ids[i]] = dx_sub[(size_t)k * ns + i];
    return 0;
  }
  // ---- the rest of the state ----
  // Lambda = Asum - Asum Pss+ Asum ;  P -= G Lambda G^T ;  dx_k = G u_k     (G = P0[:, s] in Y)
  HIPCHK(ovp_launch_gemm4(0, 0, ns, ns, ns, c->pl_Asum, ld, Pss_new, ld, c->W1, ld, 0, 0, s));
  HIPCHK(ovp_launch_gemm4(0, 0, ns, ns, ns, c->W1, ld, c->pl_Asum, ld, c->T, ld, 0, 1, s));
  HIPCHK(ovp_launch_mat_sub(c->pl_Asum, c->T, c->T, ns, ns, ld, s));
  HIPCHK(ovp_launch_gather_cols(c->P, ld, d_ids, n, ns, c->Y, ld, s));
  if (dx_planes && NP > 0) {
    // rows = planes: DX (NP x n) = U (NP x ns) G^T
    HIPCHK(ovp_launch_gemm4(0, 1, NP, n, ns, c->pl_U, ld, c->Y, ld, c->Lt, ld, 0, 0, s));
    HIPCHK(hipMemcpy2DAsync(dx_planes, sizeof(double) * n, c->Lt, sizeof(double) * ld, sizeof(double) * n, NP, hipMemcpyDeviceToHost, s));
  }
  HIPCHK(ovp_launch_gemm4(0, 0, n, ns, ns, c->Y, ld, c->T, ld, c->W1, ld, 0, 0, s));
  HIPCHK(ovp_launch_gemm4(0, 1, n, n, ns, c->W1, ld, c->Y, ld, c->L, ld, 0, 1, s));
  HIPCHK(ovp_launch_sub_sym(c->P, c->L, n, ld, s));
  HIPCHK(hipStreamSynchronize(s));
  if (dx_planes)  // the involved entries straight from the loop (the product above agrees with them to rounding)
    for (int k = 0; k < NP; ++k)
      for (int i = 0; i < ns; ++i) dx_planes[(size_t)k * n + ids[i]] = dx_sub[(size_t)k * ns + i];
  return 0;
}

extern "C" int ovp_msckf_plane_update(ovp_ctx* c, const ovp_update_opts* o, const ovp_plane_batch* pb, double* dx_planes,
                                      uint8_t* plane_ok, double* plane_chi2, int* plane_dof, uint8_t* feat_used) {
  if (!c || !o || !pb || pb->n_planes < 0) return OVP_E_ARG;
  if (!c->have_state || !c->have_cov || !c->have_batch) return OVP_E_STATE;
  if (c->h_n_meas.empty() && c->n_feats > 0) return OVP_E_STATE;  // needs ovp_batch_upload (host copy of the layout)
  const double t_entry = c->pl_sub_active ? c->pl_t_entry : host_now_ms();
  c->have_factor = false;
  const int n = c->n, ld = c->ld, F = c->n_feats, NP = pb->n_planes, M = c->max_meas;
  // skip_plane_used is an option of the POINT update that follows; the plane loop itself produces the mask
  ovp_update_opts o_local = *o;
  o_local.skip_plane_used = 0;
  o = &o_local;
  c->pl_used_valid = false;
  int rcu = ensure_pl_used(c);
  if (rcu) return rcu;
  const bool natural_order = getenv("OVP_PL_NATURAL_ORDER") != nullptr;  // A/B: the loop on all n columns in the state's order
  if (!c->pl_sub_active && NP > 0 && (n > ovp_chol2_max_n() || !natural_order))
    return plane_update_ordered(c, o, pb, dx_planes, plane_ok, plane_chi2, plane_dof, feat_used);
  if (n > ovp_chol2_max_n()) return OVP_E_CAPACITY;  // (plane_update_ordered hands over a sub-state the factorization can take)
  if (feat_used) memset(feat_used, 0, (size_t)F);
  for (int pl = 0; pl < NP; ++pl) {
    if (plane_ok) plane_ok[pl] = 0;
    if (plane_chi2) plane_chi2[pl] = 0.0;
    if (plane_dof) plane_dof[pl] = 0;
  }
  if (dx_planes && NP > 0) memset(dx_planes, 0, sizeof(double) * (size_t)n * NP);
  if (NP == 0) {  // a frame without planes: nothing is consumed, and a point update with skip_plane_used may follow
    if (F) HIPCHK(hipMemsetAsync(c->pl_used, 0, (size_t)F, c->stream));
    c->h_pl_used.assign((size_t)F, 0);
    c->pl_used_valid = true;
    return 0;
  }
  for (int k = 0; k < NP; ++k)
    if (pb->plane_state_id[k] >= 0 && pb->plane_state_id[k] + 3 > n) return OVP_E_ARG;
  const int n_slam = pb->n_slam > 0 ? pb->n_slam : 0;
  if (n_slam > 0 && (!pb->slam_plane || !pb->slam_state_id || !pb->slam_p || !pb->slam_p_fej)) return OVP_E_ARG;
  for (int q = 0; q < n_slam; ++q)
    if (pb->slam_state_id[q] < 0 || pb->slam_state_id[q] + 3 > n || pb->slam_plane[q] < 1 || pb->slam_plane[q] > NP) return OVP_E_ARG;
  int rc = fill_feat_params(c, o);
  if (rc) return rc;
  ovp::FeatParams fp = c->fp;
  fp.skip = nullptr;
  fp.range_lo = 0;  // the plane loop always walks the whole batch
  fp.range_hi = 0x7fffffff;
  // ---- what does not depend on the grouping goes to the device first: the fills and chol(P) (~70 us) run while the host sorts the
  // features by plane and builds the per-plane tables (~40 us at config 3, during which the stream used to be idle) ----
  const size_t res_bytes = sizeof(double) * (4 * (size_t)NP + (size_t)n * NP) + (size_t)F + 64 + 256;  // (+ flags and sequence word)
  rc = plane_buffers(c, NP);  // shared with the first generation: pl_res, pl_dx, pl_cst, pl_An, ...
  if (rc) return rc;
  rc = plane2_buffers(c, NP, 0, res_bytes);
  if (rc) return rc;
  hipStream_t s = c->stream;
  const double t_first = host_now_ms();
  const size_t tstride = (size_t)(c->n_max + 1) * ld;
  {
    // results, per-plane corrections, used-feature mask (rounded up to whole words: the buffer is f_max + 64 bytes), flags,
    // [0] current T buffer + [1..2] factor bookkeeping (PlaneSolve::cond), half 0 of T (sum of the accepted L0^T A L0): one launch
    // both halves of T: a plane writes its candidate only inside its leading block, the rest of either half must read as zero;
    // the packed factor / inverted diagonal blocks behind the loop start out as the identity for the same reason
    const int ntn = (n + 15) / 16;
    void* zp[8] = {c->pl_res, c->pl_dx, c->pl_used, c->flags, c->pl_cur, c->pl_Tbuf, c->Ltp, c->Dinv};
    const size_t zb[8] = {sizeof(double) * 4 * NP, sizeof(double) * (size_t)n * NP, ((size_t)F + 3) & ~(size_t)3, sizeof(int) * 4,
                          3 * sizeof(int), sizeof(double) * (tstride + (size_t)n * ld),
                          c->pl_sub_active ? sizeof(double) * 256 * (size_t)(ntn * (ntn + 1) / 2) : 0,
                          c->pl_sub_active ? sizeof(double) * 256 * (size_t)ntn : 0};
    const int zpat[8] = {0, 0, 0, 0, 0, 0, 2, 1};
    HIPCHK(ovp_launch_fill_regions(zp, zb, zpat, 8, ntn, s));
  }
  if (c->pl_ktimer) {  // [0 | 1] = the whole loop on the device clock (first launch .. covariance product), then a pair per plane
    while (c->pl_ev_loop.size() < 2) {
      hipEvent_t e;
      HIPCHK(hipEventCreate(&e));
      c->pl_ev_loop.push_back(e);
    }
    if (!c->pl_sub_active) HIPCHK(hipEventRecord(c->pl_ev_loop[0], s));  // (plane_update_ordered: in front of its permutation)
  }
  // (a cheap look at the batch first: when no plane can qualify - update/UpdaterMSCKF.cpp:316-317, 384-396 - nothing below needs the
  // factor, and a singular prior must not fail a call that has nothing to update)
  bool any_candidate = false;
  {
    std::vector<int> cnt((size_t)NP + 1, 0);
    for (int f = 0; f < F; ++f) {
      const int pf = pb->plane_of_feat[f];
      if (pf >= 1 && pf <= NP && c->h_n_meas[f] >= 2) ++cnt[pf];
    }
    for (int q = 0; q < n_slam; ++q) ++cnt[pb->slam_plane[q]];
    for (int pl = 1; pl <= NP && !any_candidate; ++pl)
      any_candidate = cnt[pl] >= (pb->plane_state_id[pl - 1] >= 0 ? 1 : 4);
  }
  // L0 = chol(P), dense lower triangular in c->L.  Nothing needs it before the first plane's W = A L0, so it runs on the side
  // stream beside that plane's rows / Gram pair / assembly (round 5; one workgroup - the small kernels of the front end leave it a
  // CU on every XCD) and is joined in front of that product (A/B in round 5: config 3 2.776 -> 2.719 ms).
  // (every way out of this function behind the fork - HIPCHK returns included - makes the loop's stream wait for the side stream:
  // the next call must not race a factorization that is still writing c->L / c->flags)
  struct ForkGuard {
    hipStream_t s;
    hipEvent_t ev_join;
    hipStream_t side;
    bool forked;
    ~ForkGuard() {
      if (!forked) return;
      (void)hipEventRecord(ev_join, side);  // (a second record behind whatever the side stream got: harmless when the first one made it)
      (void)hipStreamWaitEvent(s, ev_join, 0);
    }
  } fork_guard{s, c->ev_join, c->stream2, false};
  bool& chol_forked = fork_guard.forked;
  if (any_candidate) {
    if (s == c->stream) {
      HIPCHK(hipEventRecord(c->ev_fork, s));
      HIPCHK(hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
      chol_forked = true;
      rc = chol_of_P(c, c->stream2);
      if (rc) return rc;
      HIPCHK(hipEventRecord(c->ev_join, c->stream2));
    } else {
      rc = chol_of_P(c, s);
      if (rc) return rc;
    }
  }
  auto join_chol = [&]() -> hipError_t {
    if (!chol_forked) return hipSuccess;
    chol_forked = false;
    return hipStreamWaitEvent(s, c->ev_join, 0);
  };
  // a refusal from here on: chol(P) has run - a flag it may have raised (singular prior) must not outlive the call
  auto bail = [&](int code) {
    (void)join_chol();
    (void)hipMemsetAsync(c->flags, 0, sizeof(int) * 4, s);
    return code;
  };
  // ---- host-side grouping (update/UpdaterMSCKF.cpp:204-229) ----
  std::vector<PlaneJobH> jobs;
  std::vector<int> featlist;
  std::vector<int> perms;  // per job: n entries
  const int ncal = (o->do_calib_camera_pose ? 6 : 0) + (o->do_calib_camera_intrinsics ? 8 : 0);
  // features bucketed by plane in one pass, batch order kept (a scan of the whole batch per plane was 0.4 ms of host time in
  // front of the first launch at 8000 features x 50 planes)
  std::vector<int> bucket_start((size_t)NP + 2, 0), bucket((size_t)(F > 0 ? F : 1));
  for (int f = 0; f < F; ++f) {
    const int pf = pb->plane_of_feat[f];
    if (pf >= 1 && pf <= NP) ++bucket_start[pf + 1];
  }
  for (int pl = 1; pl <= NP + 1; ++pl) bucket_start[pl] += bucket_start[pl - 1];
  {
    std::vector<int> fill(bucket_start.begin(), bucket_start.end());
    for (int f = 0; f < F; ++f) {
      const int pf = pb->plane_of_feat[f];
      if (pf >= 1 && pf <= NP) bucket[fill[pf]++] = f;
    }
  }
  for (int pl = 0; pl < NP; ++pl) {
    PlaneJobH j;
    j.pl = pl;
    j.start = (int)featlist.size();
    j.nf = 0;
    j.rows_total = 0;
    j.rows_live = 0;
    j.sid = pb->plane_state_id[pl];
    j.in_state = j.sid >= 0;
    unsigned long long seen = 0ull;
    for (int bi = bucket_start[pl + 1]; bi < bucket_start[pl + 2]; ++bi) {
      const int f = bucket[bi];
      const int m = c->h_n_meas[f];
      if (m < 2) continue;
      if (m > OVP_MAX_MEAS_DEV) return bail(OVP_E_CAPACITY);  // 2m bearing rows = one wavefront (the constraint row is wave-uniform)
      featlist.push_back(f);
      j.nf++;
      j.rows_total += 3 * m - 3;
      j.rows_live += 2 * m - 2;  // the m identical constraint rows are one direction (k_chol2 gate)
      for (int k = 0; k < m; ++k) seen |= 1ull << c->h_clone_idx[(size_t)f * M + k];
    }
    int ns_pl = 0;  // SLAM landmarks on this (out-of-state) plane: one row and three involved columns each
    if (!j.in_state)
      for (int q = 0; q < n_slam; ++q)
        if (pb->slam_plane[q] == pl + 1) ++ns_pl;
    if (ns_pl > PA_MAXQ) return bail(OVP_E_CAPACITY);
    j.ns_pl = ns_pl;
    if (j.nf == 0 || (!j.in_state && j.nf + ns_pl < 4)) {  // update/UpdaterMSCKF.cpp:316-317,384-396
      featlist.resize(j.start);
      continue;
    }
    j.rows_total += ns_pl;
    j.rows_live += ns_pl;
    const int c_ref = 6 * __builtin_popcountll(seen) + ncal + 3 * ns_pl;
    const int rows_c = j.rows_total > c_ref ? c_ref : j.rows_total;  // UpdaterPlane::measurement_compress_inplace
    j.rows_u = j.in_state ? rows_c : rows_c - 3;
    j.n_involved = c_ref + (j.in_state ? 3 : 0);
    if (!j.in_state) j.rows_total -= 3;
    if (!j.in_state) j.rows_live -= 3;
    if (j.rows_u < 1) {
      featlist.resize(j.start);
      continue;
    }
    j.thr = o->chi2_multiplier * ovp_chi2_quantile_095(j.rows_u);
    // order of the involved columns in the normalised Gram: everything that is not a clone first, the clones last (a rank
    // deficiency - gauge freedom, planar scene - then shows up in the trailing pivots, k_chol2 mode 2)
    {
      std::vector<int> perm(n, -1);
      int pos = 0;
      std::vector<char> inv(n, 0);
      if (o->do_calib_camera_pose)
        for (int k = 0; k < 6; ++k) inv[c->calib_id + k] = 1;
      if (o->do_calib_camera_intrinsics)
        for (int k = 0; k < 8; ++k) inv[c->intr_id + k] = 1;
      if (j.in_state)
        for (int k = 0; k < 3; ++k) inv[j.sid + k] = 1;
      if (!j.in_state)
        for (int q = 0; q < n_slam; ++q)
          if (pb->slam_plane[q] == pl + 1)
            for (int k = 0; k < 3; ++k) inv[pb->slam_state_id[q] + k] = 1;
      for (int col = 0; col < n; ++col)
        if (inv[col]) perm[col] = pos++;
      for (int ci = 0; ci < (int)c->h_clone_id.size(); ++ci)
        if ((seen >> ci) & 1ull)
          for (int k = 0; k < 6; ++k) {
            const int col = c->h_clone_id[ci] + k;
            if (col >= 0 && col < n && perm[col] < 0) perm[col] = pos++;
          }
      j.n_inv_cols = pos;
      perms.insert(perms.end(), perm.begin(), perm.end());
    }
    jobs.push_back(j);
  }
  const int NJ = (int)jobs.size();
  if (NJ == 0 && any_candidate) {  // chol(P)'s verdict concerns nobody
    HIPCHK(join_chol());
    HIPCHK(hipMemsetAsync(c->flags, 0, sizeof(int) * 4, s));
  }
  // ---- staging layout: ints [featlist | sid NP | perms NJ*n | slam_plane | slam_id], doubles [cp | cp_fej | slam_p | slam_p_fej] ----
  const size_t n_int = featlist.size() + (size_t)NP + perms.size() + 2 * (size_t)n_slam;
  const size_t int_bytes = ((n_int * sizeof(int) + 15) / 16) * 16;
  const size_t n_dbl = 6 * (size_t)NP + 6 * (size_t)n_slam;
  const size_t stage_bytes = int_bytes + n_dbl * sizeof(double);
  rc = plane2_buffers(c, NP, stage_bytes, res_bytes);  // (grows the staging block when this frame needs more)
  if (rc) return rc;
  int* hi = (int*)c->pl_hstage;
  double* hd = (double*)((char*)c->pl_hstage + int_bytes);
  int* di = (int*)c->pl_dstage;
  double* dd = (double*)((char*)c->pl_dstage + int_bytes);
  size_t io = 0;
  const size_t o_feat = io;
  memcpy(hi + io, featlist.data(), sizeof(int) * featlist.size());
  io += featlist.size();
  const size_t o_sid = io;
  memcpy(hi + io, pb->plane_state_id, sizeof(int) * NP);
  io += NP;
  const size_t o_perm = io;
  if (!perms.empty()) memcpy(hi + io, perms.data(), sizeof(int) * perms.size());
  io += perms.size();
  const size_t o_spl = io;
  if (n_slam) memcpy(hi + io, pb->slam_plane, sizeof(int) * n_slam);
  io += n_slam;
  const size_t o_sidx = io;
  if (n_slam) memcpy(hi + io, pb->slam_state_id, sizeof(int) * n_slam);
  io += n_slam;
  memcpy(hd, pb->cp, sizeof(double) * 3 * NP);
  memcpy(hd + 3 * NP, pb->cp_fej, sizeof(double) * 3 * NP);
  if (n_slam) {
    memcpy(hd + 6 * NP, pb->slam_p, sizeof(double) * 3 * n_slam);
    memcpy(hd + 6 * NP + 3 * n_slam, pb->slam_p_fej, sizeof(double) * 3 * n_slam);
  }
  HIPCHK(hipMemcpyAsync(c->pl_dstage, c->pl_hstage, stage_bytes, hipMemcpyHostToDevice, s));
  const int* d_feat = di + o_feat;
  const int* d_sid = di + o_sid;
  const int* d_perm = di + o_perm;
  const int* d_spl = di + o_spl;
  const int* d_sidx = di + o_sidx;
  double* d_cp = dd;
  double* d_cpfej = dd + 3 * NP;
  double* d_slam_p = dd + 6 * NP;
  double* d_slam_pfej = dd + 6 * NP + 3 * n_slam;
  const double white_c = 1.0 / o->sigma_constraint;
  // weight of the expected energy of the rounding-decided rows in the gate statistic (k_chol2.hip); OVP_PL_NOISE_SCALE overrides the
  // calibrated constant for the study that produced it (tools/plane_gate_agreement.py --fit)
  double noise_scale = OVP_PLANE_NOISE_KAPPA;
  if (const char* ns_env = getenv("OVP_PL_NOISE_SCALE")) noise_scale = atof(ns_env);  // (read per call)
  for (int jn = 0; jn < NJ; ++jn) {
    const PlaneJobH& j = jobs[jn];
    // leading block this plane's products and factorization run on (plane_update_ordered): every column involved so far
    const int nk = c->pl_sub_active ? c->pl_nl[j.pl] : n;
    // (1) per-feature rows
    ovp::PlaneParams pp;
    pp.feat_list = d_feat + j.start;
    pp.n_local = j.nf;
    pp.plane = j.pl;
    pp.in_state = j.in_state;
    pp.plane_sid = j.sid;
    pp.white_c = white_c;
    pp.cp = d_cp;
    pp.cp_fej = d_cpfej;
    pp.cst = c->pl_cst;
    ovp::FeatParams fpl = fp;
    fpl.n = nk;
    fpl.P = c->P;
    HIPCHK(ovp_launch_plane_feat(&fpl, &pp, j.nf, s));
    // (2) Gram products
    const int chunks = (2 * j.nf + c->rows_per_chunk - 1) / c->rows_per_chunk;
    int nsplit = 1;
    HIPCHK(ovp_launch_gram_pair(c->rec, fp.n_clones, j.nf, c->rows_per_chunk, chunks, c->gramS, c->G, 3 * j.nf, c->ldg, nk + 4,
                                c->n_split, c->part, &nsplit, s));
    // (3) pair on the state columns, normalised Gram, residual energy
    ovp::PlaneAsm pa;
    memset(&pa, 0, sizeof(pa));
    pa.gramS = c->gramS;
    pa.n_clones = fp.n_clones;
    pa.n_chunks = chunks;
    pa.part = c->part;
    pa.n_split = nsplit;
    {
      const int nt16 = (nk + 4 + 15) / 16;
      pa.ntile = nt16 * (nt16 + 1) / 2;
    }
    pa.colmap = c->colmap;
    pa.n = nk;
    pa.plane_sid = j.sid;
    pa.in_state = j.in_state;
    pa.cst = c->pl_cst;
    pa.nf = j.nf;
    pa.n_slam = j.in_state ? 0 : n_slam;
    pa.plane1 = j.pl + 1;
    pa.slam_plane = d_spl;
    pa.slam_id = d_sidx;
    pa.slam_p = d_slam_p;
    pa.slam_p_fej = d_slam_pfej;
    pa.cp = d_cp + 3 * j.pl;
    pa.cp_fej = d_cpfej + 3 * j.pl;
    pa.white_c = white_c;
    pa.do_fej = fp.do_fej;
    pa.Ab = c->Ab;
    pa.lda = ld;
    pa.perm = d_perm + (size_t)jn * n;
    pa.An = c->pl_An;
    pa.ldn = ld;
    pa.bn = c->pl_bn;
    pa.eps = 1e-12;
    pa.scal = c->pl_scal;
    HIPCHK(ovp_launch_plane_assemble2(&pa, s));
    // (4) W = A L0 ;  T_try = T_cur + L0^T W ;  c = L0^T b
    HIPCHK(join_chol());  // (first plane: L0 comes from the side stream)
    HIPCHK(ovp_launch_gemm4(0, 0, nk, nk, nk, c->Ab, ld, c->L, ld, c->W1, ld, 0, 0, s));
    HIPCHK(ovp_launch_plane_dT(nk, c->L, ld, c->W1, c->Ab + (size_t)nk * ld, c->pl_Tbuf, tstride, c->pl_cur, c->pl_crow, s));
    // (5) both factorizations, gate, solve, commit
    ovp::Chol2Job j0, j1;
    memset(&j0, 0, sizeof(j0));
    memset(&j1, 0, sizeof(j1));
    j0.A = c->pl_Tbuf;
    j0.sel = c->pl_cur;
    j0.sel_xor = 1;
    j0.sel_stride = tstride;
    j0.n = nk;
    j0.ld = ld;
    j0.add_identity = 1;
    j0.mode = 1;
    j0.brow = c->pl_crow;
    j0.flag = c->flags;
    j1.A = c->pl_An;
    j1.n = j.n_inv_cols;
    j1.ld = ld;
    j1.add_identity = 0;
    j1.mode = 2;
    j1.brow = c->pl_bn;
    j1.flag = c->flags + 2;
    j1.piv_floor = 1e-5;
    ovp::PlaneSolve ps;
    memset(&ps, 0, sizeof(ps));
    ps.scal = c->pl_scal;
    ps.range_done = c->pl_range_done;
    ps.seq = ++c->pl_seq;
    {
      // The update part on two workgroups: tile columns < h and the rest (k_chol2.hip).  Measured (r03, A/B in one call): at 16 tile
      // columns (N = 240) nothing is gained (2.91 against 2.81 ms per config-3 plane loop for h = 5 .. 8: exports + a second gate
      // hand-over cost what the second CU's f64 pipe gives), so one workgroup stays the default there; from 17 tile columns on
      // (N > 255) the tile registers of one workgroup spill and the split wins (config 4, N = 285: 7.91 ms for h = 5 or 6, 8.10 for
      // 7 or 8, 8.69 unsplit).  OVP_C2_SPLIT: 0 = never, h = forced.
      const char* split_s = getenv("OVP_C2_SPLIT");  // (read per call: the tests switch it)
      const int split_env = split_s ? atoi(split_s) : -1;
      const int nb = nk + 1, ntb = (nb + 15) / 16;
      const int nst = (nb % 16 == 1) ? ntb - 1 : ntb;  // a border row alone in its tile row takes no step
      int h = ntb >= 17 ? nst / 3 : 0;  // part B also runs the back half of the chain: 5 - 6 of 18 steps measured best (7.91 ms per
                                        // config-4 plane loop against 8.10 for 7 or 8 and 8.69 unsplit)
      if (split_env >= 0) h = split_env < ntb - 1 ? split_env : 0;
      if (h > 9) h = 9;  // pl_xbuf holds nine exported steps
      j0.split_h = h;
      j0.xbuf = c->pl_xbuf;
      j0.xflag = c->pl_xflag;
      j0.xseq = ps.seq;
      ps.xzz = c->pl_xy;
      ps.xy = c->pl_xy + 16;
      ps.xsync = c->pl_xflag + 32;
    }
    ps.thr = j.thr;
    ps.rows_live = j.rows_live;
    ps.rows_u = j.rows_u;
    ps.n_involved = j.n_inv_cols;
    ps.force = pb->force_decision ? (int)pb->force_decision[j.pl] : -1;
    ps.noise_scale = noise_scale;
    ps.tol_strict = 1e-5;
    ps.tol_loose = 1e-5;
    ps.res_out = c->pl_res + 4 * j.pl;
    ps.L0 = c->L;
    ps.ld0 = ld;
    ps.n_full = n;
    ps.dx_out = c->pl_dx + (size_t)j.pl * n;
    ps.dx_last = c->pl_dxlast;
    ps.cur = c->pl_cur;
    // The covariance product behind the loop needs the factor of the last ACCEPTED T.  The last few planes leave theirs behind when
    // they are accepted (~8 us of stores each); if one of them stays the last accepted plane, the k_tilechol behind the loop
    // (94 us at N = 240) finds nothing to do.  Which plane that is, is decided on the device.
    const int emit_last = 4;
    ps.cond = c->pl_cur + 1;
    ps.seq_plane = jn + 1;
    ps.emit = (jn >= NJ - emit_last) ? 1 : 0;
    ps.Lpack = c->Ltp;
    ps.Dinv = c->Dinv;
    ps.feat_list = d_feat + j.start;
    ps.n_feat_local = j.nf;
    ps.feat_used = c->pl_used;
    ps.clone_R = c->clone_R;
    ps.clone_p = c->clone_p;
    ps.clone_id = c->clone_id;
    ps.n_clones = fp.n_clones;
    ps.cal = c->cal;
    ps.calib_id = o->do_calib_camera_pose ? c->calib_id : -1;
    ps.intr_id = o->do_calib_camera_intrinsics ? c->intr_id : -1;
    ps.cp = d_cp;
    ps.plane_sid = d_sid;
    ps.n_planes = NP;
    ps.n_slam = n_slam;
    ps.slam_id = d_sidx;
    ps.slam_p = d_slam_p;
    if (c->pl_ktimer == 1) {
      while ((int)c->pl_ev.size() < 2 * (jn + 1)) {
        hipEvent_t e;
        HIPCHK(hipEventCreate(&e));
        c->pl_ev.push_back(e);
      }
      HIPCHK(hipEventRecord(c->pl_ev[2 * jn], s));
    }
    // diagnostics: cycle stamps of the last plane's launch - only a library whose k_chol2 was compiled with them writes any
    // (tools/build_c2_stamps.sh; the product build leaves them out: their tests cost 1 us per launch)
    static const bool pl_stamps = getenv("OVP_PL_STAMPS") != nullptr && ovp_chol2_stamps_compiled();
    static long long* d_stamps = nullptr;
    if (pl_stamps) {
      if (!d_stamps) HIPCHK(hipMalloc((void**)&d_stamps, sizeof(long long) * 2 * 16 * 32));
      j0.stamps = d_stamps;
    }
    HIPCHK(ovp_launch_chol2(&j0, &j1, &ps, s));
    if (c->pl_ktimer == 1) HIPCHK(hipEventRecord(c->pl_ev[2 * jn + 1], s));
    if (c->pl_sub_rest)
      HIPCHK(ovp_launch_plane_sub_accum(c->pl_res + 4 * j.pl, c->Ab, c->pl_Asum, c->pl_dx + (size_t)j.pl * n,
                                        c->pl_U + (size_t)j.pl * ld, nk, ld, s));
    if (pl_stamps && jn == NJ - 1) {
      long long h[2 * 16 * 32];
      HIPCHK(hipStreamSynchronize(s));
      HIPCHK(hipMemcpy(h, d_stamps, sizeof(h), hipMemcpyDeviceToHost));
      const int ntb = (nk + 1 + 15) / 16;
      const long long* e = h + (ntb + 1) * 16;
      fprintf(stderr, "[plane tail, cycles] factor %lld | gate %lld | back substitution %lld | dx = L0 y %lld | commit %lld\n",
              e[0] - h[0], e[1] - e[0], e[2] - e[1], e[3] - e[2], e[4] - e[3]);
      if (atoi(getenv("OVP_PL_STAMPS")) >= 2) {
        // per step, both parts of a split factorization, relative to part A's first stamp: elimination wave 0 [start | column
        // there | eliminated | signalled], tile wave 0 [start | panel there | next column updated | published | step done]
        const long long t0 = h[0];
        for (int part = 0; part < (j0.split_h > 0 ? 2 : 1); ++part) {
          const long long* hp = h + part * 16 * 32;
          fprintf(stderr, " part %c: prologue stamps %lld %lld %lld\n", part ? 'B' : 'A', hp[13] - t0, hp[14] - t0, hp[15] - t0);
          for (int k = 0; k < ntb; ++k) {
            const long long* q = hp + k * 16;
            if (!q[0] && !q[8]) continue;
            // arrival word of the elimination step: what was missing of {column, panel, trailing} when wave 0 first looked (0 = there)
            const long long aw = q[4] - 500500500;
            const long long am = llround((double)aw / 1e6), ar = aw - am * 1000000, ap = llround((double)ar / 1e3), at = ar - ap * 1000;
            fprintf(stderr, "  k=%2d E %7lld %7lld %7lld %7lld [%3lld %3lld %3lld] | T %7lld %7lld %7lld %7lld %7lld | T7 %7lld %7lld %7lld\n", k,
                    q[0] - t0, q[1] - t0, q[2] - t0, q[3] - t0, am, ap, at, q[8] - t0, q[9] - t0, q[10] - t0, q[11] - t0, q[12] - t0,
                    q[7] - t0, q[5] - t0, q[6] - t0);
            if (k >= 2 && atoi(getenv("OVP_PL_STAMPS")) >= 3) fprintf(stderr, "        T own tiles final %7lld, column k+1 taken %7lld\n", q[13] - t0, q[14] - t0);
          }
          const long long* m = hp + (ntb + 1) * 16;
          fprintf(stderr, "  tail: factor done %lld, gate %lld, backsolve %lld, dx %lld, commit %lld\n", m[0] - t0, m[1] - t0, m[2] - t0,
                  m[3] - t0, m[4] - t0);
        }
      }
      HIPCHK(hipMemset(d_stamps, 0, sizeof(long long) * 2 * 16 * 32));
    }
  }
  // ---- the covariance, once:  P = L0 T^-1 L0^T = V^T V,  V = Lt^-1 L0^T ----
  bool factor_enqueued = false;
  if (NJ > 0) {
    {
      // chol of the accepted T (+ I) unless the last accepted plane left its factor behind; the second-generation kernel reads the
      // current half of the double buffer itself (Chol2Job::sel) - no copy into c->T in front of it
      if (n <= ovp_chol2_max_n() + 1) {
        ovp::Chol2Job jt;
        memset(&jt, 0, sizeof(jt));
        jt.A = c->pl_Tbuf;
        jt.sel = c->pl_cur;
        jt.sel_xor = 0;
        jt.sel_stride = tstride;
        jt.n = n;
        jt.ld = ld;
        jt.add_identity = 1;
        jt.mode = 0;
        jt.flag = c->flags;
        jt.Lpack = c->Ltp;
        jt.Dinv_out = c->Dinv;
        jt.skip_cond = c->pl_cur + 1;
        HIPCHK(ovp_launch_chol2(&jt, nullptr, nullptr, s));
      } else {
        HIPCHK(ovp_launch_select_copy(c->T, c->pl_Tbuf, tstride, c->pl_cur, n, ld, 1, s));
        HIPCHK(chol_of_T(c, c->T, n, ld, 1, c->pl_cur + 1, s));
      }
    }
    HIPCHK(ovp_launch_fwdsub(c->Ltp, c->Dinv, c->L, c->Y, n, ld, 0, s));
    HIPCHK(ovp_launch_gemm4c(1, 0, n, n, n, c->Y, ld, c->Y, ld, c->P, ld, 0, 1, c->flags, s));
    // back into the state's own column order (unless a factorization failed: the resident P stays), and the factor of the
    // covariance just formed for the point update behind the loop (P = V^T V: M = V^T, rows in state order) - one launch for both
    const bool keep_factor = getenv("OVP_NO_KEPT_FACTOR") == nullptr;  // (read per call: the tests switch it)
    const bool want_factor = keep_factor && !c->pl_sub_rest && n <= OVP_TILECHOL_NMAX;
    if (want_factor && !c->Lkeep) HIPCHK(dalloc(&c->Lkeep, (size_t)c->n_max * ld));
    if (c->pl_scatter_dst) {
      HIPCHK(ovp_launch_unpermute_pair(c->P, c->Y, ld, c->pl_scatter_ids, n, c->pl_scatter_dst, want_factor ? c->Lkeep : nullptr, ld,
                                       c->flags, c->pl_boost_active ? c->boost_vec : nullptr, s));
      c->kept_boost = want_factor && c->pl_boost_active;  // Lkeep is a factor of P + diag(boost_vec): the point update on it
                                                          // takes the amounts off at its end (ekf_from_gram)
    } else if (want_factor) {
      HIPCHK(ovp_launch_factor_from_V(c->Y, ld, nullptr, n, c->Lkeep, ld, s));
    }
    factor_enqueued = want_factor;
  }
  if (c->pl_ktimer) HIPCHK(hipEventRecord(c->pl_ev_loop[1], s));
  // ---- results: one pinned block, one synchronisation ----
  double* hres = (double*)c->pl_hres;
  double* hdx = hres + 4 * (size_t)NP;
  unsigned char* hused = (unsigned char*)(hdx + (size_t)n * NP);
  // (behind `used`, each on a 64-byte line of its own: the flags, the sequence word)
  char* hflags_pub = (char*)hused + (((size_t)F + 63) & ~(size_t)63);
  volatile unsigned* hseq = (volatile unsigned*)(hflags_pub + 64);
  const unsigned seq = ++c->pl_pub_seq;
  {
    char* dbase = (char*)c->pl_hres_dev;
    auto dev_of = [&](const void* hp) { return dbase + ((const char*)hp - (const char*)c->pl_hres); };
    hipLaunchKernelGGL(k_publish_plane_results, dim3(1), dim3(1024), 0, s, c->pl_res, 4 * NP, c->pl_dx, dx_planes ? n * NP : 0, c->pl_used,
                       F, c->flags, (double*)dev_of(hres), (double*)dev_of(hdx), (unsigned char*)dev_of(hused), (int*)dev_of(hflags_pub),
                       (volatile unsigned*)dev_of((const void*)hseq), seq);
    HIPCHK(hipGetLastError());
  }
  const double t_enq = host_now_ms();
  {
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (__atomic_load_n((const unsigned*)hseq, __ATOMIC_ACQUIRE) != seq) {
      if ((++spins & 0xFFFu) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {
        HIPCHK(hipStreamSynchronize(s));  // error path: surface a fault instead of spinning forever
        if (__atomic_load_n((const unsigned*)hseq, __ATOMIC_ACQUIRE) != seq) return OVP_E_STATE;
        break;
      }
      __builtin_ia32_pause();
    }
    memcpy(c->h_flags, hflags_pub, sizeof(int) * 4);
  }
  {
    c->host_acc[0] += t_first - t_entry;
    c->host_acc[1] += t_enq - t_entry;
    c->host_acc[2] += host_now_ms() - t_enq;
    c->host_acc[3] += 1.0;
  }
  if (c->pl_ktimer) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, c->pl_ev_loop[0], c->pl_ev_loop[1]) == hipSuccess) c->host_acc[7] += ms;
  }
  if (c->pl_ktimer == 1)
    for (int jn = 0; jn < NJ; ++jn) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, c->pl_ev[2 * jn], c->pl_ev[2 * jn + 1]) == hipSuccess) {
        c->pl_ktime_ms += ms;
        c->pl_klaunches += 1;
      }
    }
  c->pl_used_valid = true;
  c->h_pl_used.assign(hused, hused + F);
  if (dx_planes) memcpy(dx_planes, hdx, sizeof(double) * (size_t)n * NP);
  if (feat_used && F) memcpy(feat_used, hused, (size_t)F);
  for (const PlaneJobH& j : jobs) {
    if (plane_ok) plane_ok[j.pl] = hres[4 * j.pl + 1] > 0.5 ? 1 : 0;
    if (plane_chi2) plane_chi2[j.pl] = hres[4 * j.pl];
    if (plane_dof) plane_dof[j.pl] = j.rows_u;
  }
  const int bad = c->h_flags[0] | c->h_flags[2];  // (the device words were cleared by the publishing kernel)
  if (bad) {
    // A failed factorization / timed-out hand-over rejects its plane and every later one before anything is committed, and the
    // covariance product behind the loop is cancelled (the resident P is the prior).  Planes accepted BEFORE the failure have
    // committed their corrections to the device tables: those no longer belong to the resident covariance - the caller must
    // upload the state again (OVP_E_STATE until then).  chol(P) itself failing (singular prior) happens in front of every plane.
    bool any_committed = false;
    for (const PlaneJobH& j : jobs) any_committed |= hres[4 * j.pl + 1] > 0.5;
    if (any_committed) c->have_state = false;
    if (bad == 1 && !any_committed && !c->pl_psd) {
      // chol(P) hit a non-positive pivot: the prior is only positive SEMI-definite.  Nothing was committed and the resident
      // covariance was not written - the same loop once more on the pivot-dropping factor (chol_of_P).
      c->pl_psd = true;
      const int rc2 = ovp_msckf_plane_update(c, o, pb, dx_planes, plane_ok, plane_chi2, plane_dof, feat_used);
      c->pl_psd = false;
      return rc2;
    }
  }
  if (bad & 2) return OVP_E_TIMEOUT;
  if (bad) return OVP_E_NOTSPD;
  c->have_factor = factor_enqueued;
  return 0;
}

// ---- UpdaterPlane::init_vio_plane core ----------------------------------------------------------
extern "C" int ovp_plane_init(ovp_ctx* c, const ovp_update_opts* o, const ovp_plane_batch* pb, double const_init_multi,
                              double const_init_chi2, double* dx_planes, int dx_stride, uint8_t* plane_ok, double* plane_chi2,
                              int* plane_dof, int* new_ids, double* cp_new, uint8_t* feat_used) {
  drop_kept_factor(c);  // (writes the covariance: a kept factor no longer belongs to it)
  if (!c || !o || !pb || pb->n_planes < 0) return OVP_E_ARG;
  if (!c->have_state || !c->have_cov || !c->have_batch) return OVP_E_STATE;
  if (c->h_n_meas.empty() && c->n_feats > 0) return OVP_E_STATE;
  const int ld = c->ld, F = c->n_feats, NP = pb->n_planes, M = c->max_meas;
  if (feat_used) memset(feat_used, 0, (size_t)F);
  for (int pl = 0; pl < NP; ++pl) {
    if (plane_ok) plane_ok[pl] = 0;
    if (plane_chi2) plane_chi2[pl] = 0.0;
    if (plane_dof) plane_dof[pl] = 0;
    if (new_ids) new_ids[pl] = -1;
    if (cp_new) memcpy(cp_new + 3 * pl, pb->cp + 3 * pl, 3 * sizeof(double));
    if (dx_planes) memset(dx_planes + (size_t)pl * dx_stride, 0, sizeof(double) * dx_stride);
  }
  if (NP == 0) return 0;
  c->pl_n_slam = 0;
  int rc = fill_feat_params(c, o);
  if (rc) return rc;
  rc = plane_buffers(c, NP);
  if (rc) return rc;
  rc = plane2_buffers(c, NP, 0, 0);  // (pl_crow: scale vector of the pivot-dropping factor, chol_of_P on a semi-definite prior)
  if (rc) return rc;
  hipStream_t s = c->stream;
  const int ncal = (o->do_calib_camera_pose ? 6 : 0) + (o->do_calib_camera_intrinsics ? 8 : 0);
  std::vector<int> sid(NP, -1);
  HIPCHK(hipMemcpyAsync(c->pl_sid, sid.data(), sizeof(int) * NP, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(c->pl_cp, pb->cp, sizeof(double) * 3 * NP, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(c->pl_cp_fej, pb->cp, sizeof(double) * 3 * NP, hipMemcpyHostToDevice, s));
  HIPCHK(hipStreamSynchronize(s));
  std::vector<double> res4(4), dxh(c->n_max), dcp(3);
  bool psd_prior = false;
  // The plane runs on the MARGINAL of the columns its rows can touch (clones and calibration: at most 6 * 32 + 14 of them) and the
  // rest of the state follows from the push-through identity, like the plane loop's sub-state (plane_update_ordered):
  // update/UpdaterPlane.cpp:296-481 has no size limit, the factorizations are those of ~80 columns instead of the state's, and the
  // marginal of a prior with an exact stochastic clone is positive definite (no second attempt).  Closed-loop session with two
  // planes, per frame: 0.198 ms against 0.216 on the whole state.  OVP_PLANE_INIT_SUB=0: the whole state (<= 288 columns; A/B, tests).
  const char* sub_env = getenv("OVP_PLANE_INIT_SUB");  // (read per call)
  const bool whole_state = sub_env && sub_env[0] == '0' && c->n <= OVP_TILECHOL_NMAX;
  std::vector<int> sub_ids, sub_pos;
  SubTables sub_t;
  int ns = 0;
  if (!whole_state) {
    const int n = c->n;
    sub_pos.assign((size_t)c->n_max, -1);
    bool bad_id = false;
    auto place = [&](int id, int sz) {
      if (id < 0 || id + sz > n) {
        bad_id = true;
        return;
      }
      for (int k = 0; k < sz; ++k)
        if (sub_pos[id + k] < 0) {
          sub_pos[id + k] = (int)sub_ids.size();
          sub_ids.push_back(id + k);
        }
    };
    for (int i = 0; i < c->fp.n_clones; ++i) place(c->h_clone_id[i], 6);
    if (o->do_calib_camera_pose) place(c->calib_id, 6);
    if (o->do_calib_camera_intrinsics) place(c->intr_id, 8);
    if (bad_id) return OVP_E_ARG;
    ns = (int)sub_ids.size();
    if (ns > OVP_TILECHOL_NMAX) return OVP_E_CAPACITY;
    rc = sub_tables_upload(c, o, sub_ids, sub_pos, &sub_t, s);
    if (rc) return rc;
    if (!c->pl_Asum) HIPCHK(dalloc(&c->pl_Asum, (size_t)c->n_max * ld));
    if (c->pl_U_cap < 1) {
      c->pl_U_cap = 8;
      HIPCHK(dalloc(&c->pl_U, (size_t)c->pl_U_cap * ld));
    }
  }
  for (int pl = 0; pl < NP; ++pl) {
    const int n = c->n;
    if ((n > OVP_TILECHOL_NMAX && !ns) || n + 3 > c->n_max) return OVP_E_CAPACITY;
    std::vector<int> featlist;
    int rows_total = 0, rows_live = 0;
    unsigned long long seen = 0ull;
    for (int f = 0; f < F; ++f) {
      if (pb->plane_of_feat[f] != pl + 1) continue;
      const int m = c->h_n_meas[f];
      if (m < 2) continue;
      if (m > OVP_MAX_MEAS_DEV) return OVP_E_CAPACITY;
      featlist.push_back(f);
      rows_total += 3 * m - 3;
      rows_live += 2 * m - 2;
      for (int k = 0; k < m; ++k) seen |= 1ull << c->h_clone_idx[(size_t)f * M + k];
    }
    const int nf = (int)featlist.size();
    if (nf < 3) continue;  // update/UpdaterPlane.cpp:303
    const int c_ref = 6 * __builtin_popcountll(seen) + ncal;
    const int rows_c = rows_total > c_ref ? c_ref : rows_total;
    if (rows_c - 3 < 1) continue;
    // the chi2 of StateHelper::initialize covers the rows that do not involve the plane, with dof = all rows (:471)
    const double thr = const_init_chi2 * ovp_chi2_quantile_095(rows_c);
    HIPCHK(hipMemcpyAsync(c->pl_featlist, featlist.data(), sizeof(int) * nf, hipMemcpyHostToDevice, s));
    ovp::FeatParams fp = c->fp;
    // A prior that is only positive SEMI-definite (an exact stochastic clone in front of the next propagation: every frame of a
    // running filter) fails chol(P) before anything is committed: the plane runs once more on the pivot-dropping factor of the
    // unit-diagonal form (chol_of_P, as the plane loop does), and so do the planes behind it.
    const int nj = ns ? ns : n;  // the size the plane's kernels run on
    SubSaved sub_sv;
    if (ns) {  // the marginal of the selection stands in for the state
      HIPCHK(ovp_launch_gather_block(c->P, ld, sub_t.d_ids, ns, c->P_tmp, ld, s));
      sub_sv = sub_enter(c, sub_t, ns, c->P_tmp);
      rc = fill_feat_params(c, o);  // (the calibration columns of the selection)
      if (rc) {
        sub_leave(c, sub_sv);
        (void)fill_feat_params(c, o);
        return rc;
      }
      fp = c->fp;
    }
    for (int attempt = 0; attempt < 2; ++attempt) {
      c->pl_psd = psd_prior;
      rc = (int)hipMemsetAsync(c->flags, 0, sizeof(int) * 4, s);
      if (!rc) rc = (int)hipMemsetAsync(c->pl_res + 4 * pl, 0, sizeof(double) * 4, s);
      if (!rc) rc = chol_of_P(c, s);
      if (!rc)
        rc = plane_job_device(c, o, fp, pl, 0, nf, 0, -1, 1.0 / (const_init_multi * o->sigma_constraint), c->L, 0, thr, rows_live - 3,
                              rows_c - 3, c_ref);
      c->pl_psd = false;
      if (!rc) rc = (int)hipMemcpyAsync(res4.data(), c->pl_res + 4 * pl, sizeof(double) * 4, hipMemcpyDeviceToHost, s);
      if (!rc) rc = (int)hipMemcpyAsync(c->h_flags, c->flags, sizeof(int) * 4, hipMemcpyDeviceToHost, s);
      if (!rc) rc = (int)hipStreamSynchronize(s);
      if (rc) break;  // (the selection's tables are taken off below before the error goes out)
      if (!c->h_flags[0] || psd_prior || nj > ovp_chol2_max_n() + 1) break;
      psd_prior = true;
    }
    if (ns) {
      sub_leave(c, sub_sv);
      const int rf = fill_feat_params(c, o);
      if (!rc) rc = rf;
      fp = c->fp;
    }
    if (rc) return rc;
    if (c->h_flags[0]) {
      (void)hipMemsetAsync(c->flags, 0, sizeof(int) * 4, s);
      return OVP_E_NOTSPD;
    }
    if (plane_chi2) plane_chi2[pl] = res4[0];
    if (plane_dof) plane_dof[pl] = rows_c;
    if (res4[1] < 0.5) continue;  // chi2 rejected: StateHelper::initialize returns false
    // accepted: P <- P+ = V^T V, append the plane, update the device tables like Type::update would
    if (ns) {
      // the selection's posterior Pss+ = V^T V; the whole state by the push-through identity (G = P[:, s], A|b = the plane's pair on s):
      //   u = b - A dx_s,  dx = G u ;  Lambda = A - A Pss+ A,  P -= G Lambda G^T
      HIPCHK(ovp_launch_gemm4(1, 0, ns, ns, ns, c->Y, ld, c->Y, ld, c->P_tmp, ld, 0, 1, s));
      HIPCHK(hipMemsetAsync(c->pl_Asum, 0, sizeof(double) * (size_t)ns * ld, s));
      HIPCHK(ovp_launch_plane_sub_accum(c->pl_res + 4 * pl, c->Ab, c->pl_Asum, c->dx, c->pl_U, ns, ld, s));  // (Asum = A from here)
      HIPCHK(ovp_launch_gemm4(0, 0, ns, ns, ns, c->pl_Asum, ld, c->P_tmp, ld, c->W1, ld, 0, 0, s));
      HIPCHK(ovp_launch_gemm4(0, 0, ns, ns, ns, c->W1, ld, c->pl_Asum, ld, c->T, ld, 0, 1, s));
      HIPCHK(ovp_launch_mat_sub(c->pl_Asum, c->T, c->T, ns, ns, ld, s));
      HIPCHK(ovp_launch_gather_cols(c->P, ld, sub_t.d_ids, n, ns, c->Y, ld, s));
      HIPCHK(ovp_launch_gemm4(0, 1, 1, n, ns, c->pl_U, ld, c->Y, ld, c->Lt, ld, 0, 0, s));
      HIPCHK(hipMemcpyAsync(c->dx, c->Lt, sizeof(double) * n, hipMemcpyDeviceToDevice, s));
      HIPCHK(ovp_launch_gemm4(0, 0, n, ns, ns, c->Y, ld, c->T, ld, c->W1, ld, 0, 0, s));
      HIPCHK(ovp_launch_gemm4(0, 1, n, n, ns, c->W1, ld, c->Y, ld, c->L, ld, 0, 1, s));
      HIPCHK(ovp_launch_sub_sym(c->P, c->L, n, ld, s));
      HIPCHK(ovp_launch_plane_init_augment(c->pl_E, c->ldg, ns, sub_t.d_ids, n, c->P, ld, c->dx, c->pl_scal + 4, s));
    } else {
      HIPCHK(ovp_launch_gemm4(1, 0, n, n, n, c->Y, ld, c->Y, ld, c->P, ld, 0, 1, s));
      HIPCHK(ovp_launch_plane_init_augment(c->pl_E, c->ldg, n, nullptr, n, c->P, ld, c->dx, c->pl_scal + 4, s));
    }
    HIPCHK(ovp_launch_plane_commit(c->pl_res + 4 * pl, nullptr /* no factor is chained here */, nullptr, n, ld, c->dx,
                                   c->pl_dx + (size_t)pl * c->n_max, c->clone_R, c->clone_p, c->clone_id, fp.n_clones, c->cal,
                                   o->do_calib_camera_pose ? c->calib_id : -1, o->do_calib_camera_intrinsics ? c->intr_id : -1,
                                   c->pl_cp, c->pl_sid, 0, s));
    HIPCHK(hipMemcpyAsync(dxh.data(), c->dx, sizeof(double) * n, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(dcp.data(), c->pl_scal + 4, sizeof(double) * 3, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    c->n = n + 3;
    if (plane_ok) plane_ok[pl] = 1;
    if (new_ids) new_ids[pl] = n;
    if (cp_new)
      for (int k = 0; k < 3; ++k) cp_new[3 * pl + k] = pb->cp[3 * pl + k] + dcp[k];
    if (dx_planes) memcpy(dx_planes + (size_t)pl * dx_stride, dxh.data(), sizeof(double) * (n < dx_stride ? n : dx_stride));
    if (feat_used)
      for (int f : featlist) feat_used[f] = 1;
  }
  return 0;
}

