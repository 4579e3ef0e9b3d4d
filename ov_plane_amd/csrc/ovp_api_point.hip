// C-ABI shim, part 2 (see ovp_ctx.h): the point update - UpdaterMSCKF::update downstream of the plane loop
// (update/UpdaterMSCKF.cpp:671-814, state/StateHelper.cpp:121-202) and StateHelper::EKFUpdate with a dense host H.
#include "ovp_ctx.h"

// Results go to the host without a copy command: the last kernel of an update writes the result block into mapped pinned
// memory and then a sequence number; ovp_msckf_fetch_results spins on that word (a hipMemcpyAsync + hipStreamSynchronize
// pair costs ~25 us of launch, blit and wake-up latency per update, this ~5).
__global__ __launch_bounds__(1024) void k_publish_results(unsigned long long* __restrict__ src,
                                                         unsigned long long* __restrict__ dst, int words,
                                                         volatile unsigned* seq_host, unsigned seq) {
  for (int i = threadIdx.x; i < words; i += 1024) dst[i] = src[i];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) *seq_host = seq;
  if (threadIdx.x < 2) src[threadIdx.x] = 0ull;  // the four flag words, cleared for the next update
}

// ---- the update step ---------------------------------------------------------------------------
// States above the tile factorization's limit (N > 288: e.g. 30 clones plus 50 landmarks).  A measurement batch never
// touches all of such a state: A = H^T H is non-zero on ns <= 288 involved columns s only.  With G = P[:, s]:
//     P+ = P - G (A - A Pss+ A) G^T ,   Pss+ = (Pss^-1 + A)^-1   (from (I + P A)^-1 = I - P+ A restricted to s),
// so the factorizations run on the ns x ns problem through the same tile kernels and the rest is three MFMA products.
static bool substate_ok(const ovp_ctx* c) { return c->n > OVP_TILECHOL_NMAX && c->sub_ns > 0 && c->sub_ns <= OVP_TILECHOL_NMAX; }

int set_substate(ovp_ctx* c, const std::vector<int>& ids) {
  c->sub_ns = 0;
  if (c->n <= OVP_TILECHOL_NMAX || ids.empty() || (int)ids.size() > OVP_TILECHOL_NMAX) return 0;
  if (!c->sub_ids) HIPCHK(hipMalloc((void**)&c->sub_ids, sizeof(int) * (OVP_TILECHOL_NMAX + 16)));
  if (!c->sub_buf) HIPCHK(dalloc(&c->sub_buf, (size_t)6 * OVP_TILECHOL_NMAX * OVP_TILECHOL_NMAX));
  HIPCHK(hipMemcpyAsync(c->sub_ids, ids.data(), sizeof(int) * ids.size(), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));  // ids is the caller's temporary
  c->sub_ns = (int)ids.size();
  return 0;
}

int chol_of_P(ovp_ctx* c, hipStream_t s) {
  const int n = c->n, ld = c->ld;
  if (c->pl_psd && n <= ovp_chol2_max_n() + 1) {
    // Positive SEMI-definite prior (plane loop, second attempt: state/StateHelper.cpp:159-187 never factors P, so the reference
    // updates such a covariance - right after StateHelper::clone the newest pose is an exact copy, :346-396).  ANY factor with
    // L0 L0^T = P serves the loop (P_k = L0 (I + L0^T A L0)^-1 L0^T is the matrix inversion lemma, no inverse of P in it): the
    // pivot-dropping Cholesky of the unit-diagonal form, L0 = D Lc with zero columns where P determines nothing.
    HIPCHK(ovp_launch_unit_diag(c->P, n, ld, c->W1, c->pl_crow, s));
    ovp::Chol2Job j;
    memset(&j, 0, sizeof(j));
    j.A = c->W1;
    j.n = n;
    j.ld = ld;
    j.mode = 0;
    j.flag = c->flags;
    j.piv_floor = 1e-12;  // regular pivots of the unit-diagonal form are >= 1 / cond (1e-8 at worst), dropped ones rounding noise
    j.Ldense = c->L;
    j.ldo = ld;
    HIPCHK(ovp_launch_chol2(&j, nullptr, nullptr, s));
    HIPCHK(ovp_launch_scale_rows(c->L, n, ld, c->pl_crow, s));
    return 0;
  }
  if (n <= ovp_chol2_max_n() + 1) {  // dense factor from the second-generation kernel
    ovp::Chol2Job j;
    memset(&j, 0, sizeof(j));
    j.A = c->P;
    j.n = n;
    j.ld = ld;
    j.mode = 0;
    j.flag = c->flags;
    j.Ldense = c->L;
    j.ldo = ld;
    return (int)ovp_launch_chol2(&j, nullptr, nullptr, s);
  }
  if (n <= OVP_TILECHOL_NMAX) return (int)ovp_launch_tilechol(c->P, c->L, nullptr, nullptr, n, ld, c->flags, 0, s);
  if (substate_ok(c)) return 0;  // the sub-state update factors Pss, not P
  return (int)ovp_launch_chol(c->P, c->L, n, ld, c->flags, 0, s);
}

// chol(T) behind an update: tile-packed factor + inverted diagonal blocks for k_fwdsub.  The second-generation kernel (k_chol2:
// fused elimination, role hand-over through LDS counters instead of workgroup barriers) took over from k_tilechol in round 3
// (the first generation serves the sizes above k_chol2's register budget).
hipError_t chol_of_T(ovp_ctx* c, const double* T, int n, int ld, int add_identity, const int* cond, hipStream_t s) {
  if (n > ovp_chol2_max_n() + 1)
    return ovp_launch_tilechol_unless(T, nullptr, c->Dinv, c->Ltp, n, ld, c->flags, add_identity, cond, s);
  return ovp_launch_chol2_packed(T, c->Dinv, c->Ltp, n, ld, c->flags, add_identity, cond, s);
}

static int ekf_substate(ovp_ctx* c, bool psd = false) {
  const int n = c->n, ld = c->ld, ns = c->sub_ns, lds = OVP_TILECHOL_NMAX;
  const size_t sz = (size_t)OVP_TILECHOL_NMAX * OVP_TILECHOL_NMAX;
  double *S_P = c->sub_buf, *S_A = S_P + sz, *S_L = S_A + sz, *S_W = S_L + sz, *S_T = S_W + sz, *S_Y = S_T + sz;
  hipStream_t s = c->stream;
  HIPCHK(ovp_launch_gather_block(c->P, ld, c->sub_ids, ns, S_P, lds, s));
  HIPCHK(ovp_launch_gather_block(c->Ab, ld, c->sub_ids, ns, S_A, lds, s));
  // Pss+ = Ls (I + Ls^T A Ls)^-1 Ls^T exactly as the full-state path does it
  if (psd) {
    // second attempt behind a failed chol(Pss): positive SEMI-definite prior (an exact stochastic clone) - any factor with
    // Ls Ls^T = Pss serves the identity above; the pivot-dropping Cholesky of the unit-diagonal form (cf. chol_of_P)
    HIPCHK(ovp_launch_unit_diag(S_P, ns, lds, S_W, c->dx, s));
    ovp::Chol2Job j;
    memset(&j, 0, sizeof(j));
    j.A = S_W;
    j.n = ns;
    j.ld = lds;
    j.mode = 0;
    j.flag = c->flags;
    j.piv_floor = 1e-12;
    j.Ldense = S_L;
    j.ldo = lds;
    HIPCHK(ovp_launch_chol2(&j, nullptr, nullptr, s));
    HIPCHK(ovp_launch_scale_rows(S_L, ns, lds, c->dx, s));
  } else {
    HIPCHK(ovp_launch_tilechol(S_P, S_L, nullptr, nullptr, ns, lds, c->flags, 0, s));
  }
  HIPCHK(ovp_launch_gemm4(0, 0, ns, ns, ns, S_A, lds, S_L, lds, S_W, lds, 0, 0, s));
  HIPCHK(ovp_launch_gemm4(1, 0, ns, ns, ns, S_L, lds, S_W, lds, S_T, lds, 1, 1, s));
  HIPCHK(ovp_launch_tilechol(S_T, nullptr, c->Dinv, c->Ltp, ns, lds, c->flags, 0, s));
  HIPCHK(ovp_launch_fwdsub(c->Ltp, c->Dinv, S_L, S_Y, ns, lds, 0, s));
  HIPCHK(ovp_launch_gemm4(1, 0, ns, ns, ns, S_Y, lds, S_Y, lds, S_P, lds, 0, 1, s));
  // Lambda = A - A Pss+ A
  HIPCHK(ovp_launch_gemm4(0, 0, ns, ns, ns, S_A, lds, S_P, lds, S_W, lds, 0, 0, s));
  HIPCHK(ovp_launch_gemm4(0, 0, ns, ns, ns, S_W, lds, S_A, lds, S_T, lds, 0, 1, s));
  HIPCHK(ovp_launch_mat_sub(S_A, S_T, S_T, ns, ns, lds, s));
  // P -= G Lambda G^T   (G in Y, G Lambda in W1, the product in T)
  HIPCHK(ovp_launch_gather_cols(c->P, ld, c->sub_ids, n, ns, c->Y, ld, s));
  HIPCHK(ovp_launch_gemm4(0, 0, n, ns, ns, c->Y, ld, S_T, lds, c->W1, ld, 0, 0, s));
  HIPCHK(ovp_launch_gemm4(0, 1, n, n, ns, c->W1, ld, c->Y, ld, c->T, ld, 0, 1, s));
  HIPCHK(ovp_launch_sub_sym_unless(c->P, c->T, n, ld, c->flags, s));  // (a failed factorization leaves the resident covariance alone)
  return 0;
}

int ekf_from_gram(ovp_ctx* c, bool chol_p_done_on_stream2, bool publish) {
  const int n = c->n, ld = c->ld;
  if (!chol_p_done_on_stream2) {
    int rc = chol_of_P(c, c->stream);
    if (rc) return rc;
  } else {
    if (c->need_join) HIPCHK(hipStreamWaitEvent(c->stream, c->ev_join, 0));
  }
  const double* b = c->Ab + (size_t)n * ld;
  if (n <= OVP_TILECHOL_NMAX) {
    // W1 = A L ;  T = I + L^T W1 ;  Lt = chol(T) (+ inverses of its diagonal blocks).  L: chol(P), or the dense factor the plane
    // loop left (any M with M M^T = P gives P+ = M (I + M^T A M)^-1 M^T)
    const bool kept = c->use_kept_factor;
    const double* Lf = kept ? c->Lkeep : c->L;
    // leading block of T (reversed-order factor of P, see ovp_build_gate_gram_tail): n when the factor is the plain one
    const int nl = (!kept && chol_p_done_on_stream2 && c->point_nl > 0 && c->point_nl < n) ? c->point_nl : n;
    // diagonal amounts to take off at the end: CholJob::boost of the reversed-order factor (the first point_boost_n columns), or -
    // on the factor the plane loop left, which is a factor of P + diag(boost_vec) - the loop's own (all n entries, zero where none)
    const double* boost_ptr = kept ? (c->kept_boost ? c->boost_vec : nullptr) : c->boost;
    const int boost_n = kept ? (c->kept_boost ? n : 0) : (nl < n ? c->point_boost_n : 0);
    c->kept_boost = false;
    c->use_kept_factor = false;
    c->point_nl = 0;
    c->point_boost_n = 0;
    c->have_factor = false;  // P is about to change
    HIPCHK(ovp_launch_gemm4(0, 0, n, nl, n, c->Ab, ld, Lf, ld, c->W1, ld, 0, 0, c->stream));
    HIPCHK(ovp_launch_gemm4(1, 0, nl, nl, n, Lf, ld, c->W1, ld, c->T, ld, 1, 1, c->stream));
    HIPCHK(chol_of_T(c, c->T, nl, ld, 0, nullptr, c->stream));
    // V = Lt^-1 L^T ;  P+ = V^T V ;  dx = P+ b
    HIPCHK(ovp_launch_fwdsub_lead(c->Ltp, c->Dinv, Lf, c->Y, n, ld, kept ? 1 : (nl < n ? 2 : 0), nl, c->stream));
    // (skipped on the device when a factorization failed: the resident covariance then stays what it was, OVP_E_NOTSPD)
    HIPCHK(ovp_launch_gemm4c(1, 0, n, n, n, c->Y, ld, c->Y, ld, c->P, ld, 0, 1, c->flags, c->stream));
    if (publish) {
      // the last block of the dx kernel also publishes [flags | dx] to the pinned host block (no separate launch)
      const int words = (int)((16 + sizeof(double) * (size_t)n + 7) / 8);
      c->pub_seq = ++c->seq;
      HIPCHK(ovp_launch_dx_rows_boost(c->P, n, ld, b, c->dx, c->flags + 1, c->ticket, c->res_block, c->h_res_block_dev, words,
                                      (char*)c->h_res_block_dev + ((char*)c->h_seq - (char*)c->h_res_block), c->pub_seq, boost_ptr,
                                      boost_n, c->flags, c->stream));
      c->pub_pending = true;
    } else {
      HIPCHK(ovp_launch_dx_rows_boost(c->P, n, ld, b, c->dx, c->flags + 1, nullptr, nullptr, nullptr, 0, nullptr, 0u, boost_ptr, boost_n,
                                      c->flags, c->stream));
    }
    return 0;
  }
  if (substate_ok(c)) {
    int rc = ekf_substate(c);
    if (rc) return rc;
    if (publish) {
      const int words = (int)((16 + sizeof(double) * (size_t)n + 7) / 8);
      c->pub_seq = ++c->seq;
      HIPCHK(ovp_launch_dx_rows(c->P, n, ld, b, c->dx, c->flags + 1, c->ticket, c->res_block, c->h_res_block_dev, words,
                                (char*)c->h_res_block_dev + ((char*)c->h_seq - (char*)c->h_res_block), c->pub_seq, c->stream));
      c->pub_pending = true;
    } else {
      HIPCHK(ovp_launch_dx_rows(c->P, n, ld, b, c->dx, c->flags + 1, nullptr, nullptr, nullptr, 0, nullptr, 0u, c->stream));
    }
    return 0;
  }
  // large-state fallback when the measurements touch more than 288 columns: global-memory factorization
  HIPCHK(ovp_launch_gemm(0, 0, n, n, n, c->Ab, ld, c->L, ld, c->W1, ld, 0, c->stream));
  HIPCHK(ovp_launch_gemm(1, 0, n, n, n, c->L, ld, c->W1, ld, c->T, ld, 1, c->stream));
  HIPCHK(ovp_launch_chol(c->T, c->Lt, n, ld, c->flags, 0, c->stream));
  HIPCHK(ovp_launch_trsm_right_lt(c->L, c->Lt, c->Y, n, ld, c->stream));
  HIPCHK(ovp_launch_cov_finish(c->Y, n, ld, b, c->P, ld, c->dx, c->flags + 1, c->stream));
  return 0;
}

// Update of a covariance that is positive SEMI-definite (an exact stochastic clone before the next propagation, a zero-variance
// prior): P has no Cholesky factor, but the reference's own form needs none (state/StateHelper.cpp:159-187):
//   P+ = P - P H^T (H P H^T + I)^-1 H P,   with H := La^T, La La^T = A the (pivot-dropping) Cholesky factor of the information
// matrix of the batch - H^T H = A and H^T r = b is all the update depends on.  S = I + La^T P La is positive definite whatever P is.
// Runs after a failed chol(P) (flags[0]): the pair [A | b] is still in c->Ab, P was not touched (ovp_launch_gemm4c cancel flag).
int ekf_sform(ovp_ctx* c) {
  const int n = c->n, ld = c->ld;
  hipStream_t s = c->stream;
  if (n > OVP_TILECHOL_NMAX || n > ovp_chol2_max_n()) {
    // above the tile factorization: the sub-state update once more, on the pivot-dropping factor of the involved block
    if (!substate_ok(c) || c->sub_ns > ovp_chol2_max_n()) return OVP_E_NOTSPD;
    HIPCHK(hipMemsetAsync(c->flags, 0, sizeof(int) * 4, s));
    const int rs = ekf_substate(c, true);
    if (rs) return rs;
    HIPCHK(ovp_launch_dx_rows(c->P, n, ld, c->Ab + (size_t)n * ld, c->dx, c->flags + 1, nullptr, nullptr, nullptr, 0, nullptr, 0u, s));
    HIPCHK(hipMemcpyAsync(c->h_dx, c->dx, sizeof(double) * n, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(c->h_flags, c->flags, sizeof(int) * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipMemsetAsync(c->flags, 0, sizeof(int) * 4, s));
    return 0;
  }
  HIPCHK(hipMemsetAsync(c->flags, 0, sizeof(int) * 4, s));
  HIPCHK(hipMemsetAsync(c->W1, 0, sizeof(double) * (size_t)n * ld, s));
  HIPCHK(ovp_launch_max_diag(c->Ab, n, ld, c->smallbuf, s));
  ovp::Chol2Job j;
  memset(&j, 0, sizeof(j));
  j.A = c->Ab;
  j.n = n;
  j.ld = ld;
  j.mode = 0;
  j.flag = c->flags + 3;
  j.piv_floor = 1e-13;  // directions that carry less than 1e-13 of the largest diagonal entry count as unobserved
  j.floor_scale = c->smallbuf;
  j.Ldense = c->W1;     // La (lower triangular, zero columns where a pivot was dropped)
  j.ldo = ld;
  HIPCHK(ovp_launch_chol2(&j, nullptr, nullptr, s));
  HIPCHK(ovp_launch_gemm4(0, 0, n, n, n, c->P, ld, c->W1, ld, c->Y, ld, 0, 0, s));   // Wm = P La
  HIPCHK(ovp_launch_gemm4(1, 0, n, n, n, c->W1, ld, c->Y, ld, c->T, ld, 1, 1, s));   // S = I + La^T Wm
  HIPCHK(ovp_launch_tilechol(c->T, nullptr, c->Dinv, c->Ltp, n, ld, c->flags, 0, s));
  HIPCHK(ovp_launch_fwdsub(c->Ltp, c->Dinv, c->Y, c->L, n, ld, 1, s));               // V = Ls^-1 Wm^T
  HIPCHK(ovp_launch_gemm4(1, 0, n, n, n, c->L, ld, c->L, ld, c->T, ld, 0, 1, s));    // V^T V
  HIPCHK(ovp_launch_sub_sym(c->P, c->T, n, ld, s));
  HIPCHK(ovp_launch_dx_rows(c->P, n, ld, c->Ab + (size_t)n * ld, c->dx, c->flags + 1, nullptr, nullptr, nullptr, 0, nullptr, 0u, s));
  HIPCHK(hipMemcpyAsync(c->h_dx, c->dx, sizeof(double) * n, hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(c->h_flags, c->flags, sizeof(int) * 4, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  HIPCHK(hipMemsetAsync(c->flags, 0, sizeof(int) * 4, s));
  return 0;
}

int fill_feat_params(ovp_ctx* c, const ovp_update_opts* o);
static int ovp_build_gate_gram_tail(ovp_ctx* c, int n, int F);

extern "C" int ovp_msckf_build_gate_gram_async(ovp_ctx* c, const ovp_update_opts* o) {
  if (!c || !o) return OVP_E_ARG;
  if (!c->have_state || !c->have_cov || !c->have_batch) return OVP_E_STATE;
  if (c->fp.n_clones < 1) return OVP_E_STATE;
  const int n = c->n, F = c->n_feats;
  {
    int rc = fill_feat_params(c, o);
    if (rc) return rc;
  }
  // (the flag words were cleared by the previous ovp_msckf_fetch_results, or at creation)
  return ovp_build_gate_gram_tail(c, n, F);
}

int fill_feat_params(ovp_ctx* c, const ovp_update_opts* o) {
  const int n = c->n, F = c->n_feats;
  ovp::FeatParams& fp = c->fp;
  fp.n_feats = F;
  fp.max_meas = c->max_meas;
  fp.do_fej = o->do_fej;
  fp.calmask = (o->do_calib_camera_pose ? 0x3Fu : 0u) | (o->do_calib_camera_intrinsics ? (0xFFu << 6) : 0u);
  for (int k = 0; k < 14; ++k) {
    fp.calcol[k] = (k < 6) ? c->calib_id + k : c->intr_id + (k - 6);
    if (!((fp.calmask >> k) & 1)) fp.calcol[k] = 0;
    else if (fp.calcol[k] < 0 || fp.calcol[k] >= n) return OVP_E_ARG;
  }
  fp.white_px = 1.0 / o->sigma_px;
  fp.chi2_mult = o->chi2_multiplier;
  fp.chi2_table = c->chi2_table;
  fp.P = c->P;
  fp.n = n;
  fp.ldp = c->ld;
  fp.G = c->G;
  fp.Bscr = c->Bscr;
  fp.ldg = c->ldg;
  fp.rec = c->rec;
  // per-feature results straight into the pinned host block (same layout as res_block): they cross PCIe while K1 runs
  fp.chi2 = (double*)((char*)c->h_res_block_dev + ((char*)c->chi2 - (char*)c->res_block));
  fp.accept = (unsigned char*)c->h_res_block_dev + ((char*)c->accept - (char*)c->res_block);
  fp.dbg_cycles = c->dbg_cycles;
  fp.slot = nullptr;
  fp.n_out = 0;
  fp.skip = nullptr;
  fp.range_lo = c->range_lo < 0 ? 0 : c->range_lo;
  fp.range_hi = c->range_lo < 0 ? 0x7fffffff : c->range_hi;
  if (o->skip_plane_used) {
    if (!c->pl_used_valid || !c->pl_used) return OVP_E_STATE;  // no plane update ran on this batch
    fp.skip = c->pl_used;
  }
  return 0;
}

static int build_gate_gram_tail_impl(ovp_ctx* c, int n, int F);
static int ovp_build_gate_gram_tail(ovp_ctx* c, int n, int F) {
  const int rc = build_gate_gram_tail_impl(c, n, F);
  if (rc && c->need_join) {
    // a failing exit behind the fork of chol(P): the main stream waits for the side stream before anything else is enqueued on it
    // (the regular join sits in ekf_from_gram, which a failed build never reaches)
    (void)hipEventRecord(c->ev_join, c->stream2);
    (void)hipStreamWaitEvent(c->stream, c->ev_join, 0);
    c->need_join = false;
  }
  return rc;
}
static int build_gate_gram_tail_impl(ovp_ctx* c, int n, int F) {
  ovp::FeatParams& fp = c->fp;
  // Round 5: the features that are not part of this update - consumed by an accepted plane (skip mask), outside this rank's index
  // range - no longer occupy rows of rec / G (they used to write 16 KB of zeros each, which K2 then read): the host knows both sets,
  // numbers the others in batch order and K1 reads its slot from host-mapped memory at the start of the feature wave; K2 runs on Fa
  // features.  Same sums over the same rows in the same order, the chunks / splits of K2 group fewer of them (A/B in round 5: point update 0.2445 -> 0.2375 ms at config 3).
  int Fa = F;
  {
    const bool ranged = c->range_lo >= 0;
    const bool masked = fp.skip != nullptr && c->pl_used_valid && (int)c->h_pl_used.size() == F;
    if ((ranged || masked) && F > 0) {
      const int lo = ranged ? c->range_lo : 0, hi = ranged ? (c->range_hi < F ? c->range_hi : F) : F;
      int k = 0;
      for (int f = 0; f < F; ++f) c->h_slot[f] = (f >= lo && f < hi && !(masked && c->h_pl_used[f])) ? k++ : -1;
      Fa = k;
      fp.slot = c->d_slot;
      fp.n_out = Fa;
    }
  }
  if (n > OVP_TILECHOL_NMAX) {  // the columns a point-feature batch can touch: clones, and calibration when it is estimated
    std::vector<int> ids;
    for (int cid : c->h_clone_id)
      for (int k = 0; k < 6; ++k) ids.push_back(cid + k);
    if (fp.calmask & 0x3Fu)
      for (int k = 0; k < 6; ++k) ids.push_back(c->calib_id + k);
    if (fp.calmask & (0xFFu << 6))
      for (int k = 0; k < 8; ++k) ids.push_back(c->intr_id + k);
    for (int dc : c->dense_cols) ids.push_back(dc);
    std::sort(ids.begin(), ids.end());
    ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
    int rs = set_substate(c, ids);
    if (rs) return rs;
  }
  // chol(P) does not depend on the measurements.  Default (mode 3): it rides in workgroup 0 of the fused feature kernel, on a
  // CU of its own, and is hidden behind the features; K2 follows on the main stream and nothing forks or joins.
  // Beside a one-wave-per-block K1 it must NOT run: K1 keeps every SIMD busy with two feature waves, and the CU that also
  // hosts the eight Cholesky waves finishes its feature blocks ~40 % later, which is the kernel's duration (K1 112 -> 157 us).
  // OVP_OVERLAP_MODE: 0 = no overlap, 1 = side stream beside K1, 2 = main stream after K1 with K2 beside it on the side
  // stream (the fallback when the fused kernel cannot take the batch), 3 (default) = fused.  (Also tried: chol(P) as its own
  // 160 KB-LDS launch on the side stream beside an 8-wave-workgroup K1 - the two launches did not overlap, K1 145 us.)
  // Round 5, mode 4 (default up to 1976 features): the features keep the fused kernel's shape - eight feature waves per workgroup,
  // each workgroup a CU of its own - but workgroup 0 factorizes nothing and returns at once; chol(P) runs as the second-generation
  // kernel (k_chol2, mode 0, reversed order / diagonal boost as CholJob has them) on the side stream, on a CU the features leave
  // free: 160 KB of LDS per feature workgroup keep the two off each other's SIMDs, and a round of 247 feature workgroups leaves a
  // CU on EVERY XCD (ovp_launch_feat_chol) - with 255 the side kernel's workgroup finds no CU on the XCD it is sent to and the
  // launches serialise (config 2, 2000 features: 316 against 274 us per update, measured), so above 1976 features mode 3 stays.
  // Closed-loop session (11 clones, ~100 features): msckf update 0.276 -> 0.260 ms per frame.
  const char* overlap_env = getenv("OVP_OVERLAP_MODE");  // (read per call: the tests switch it)
  int overlap_mode = overlap_env ? atoi(overlap_env) : 4;
  const bool fused_ok = c->n <= OVP_TILECHOL_NMAX && ovp_feat_chol_supported(&fp, c->n);
  if (overlap_mode == 4 && !(fused_ok && c->n <= ovp_chol2_max_n() && F <= ovp_feat_chol_side_capacity())) overlap_mode = 3;
  if (overlap_mode == 3 && !fused_ok) overlap_mode = 2;
  c->need_join = (overlap_mode == 1 || overlap_mode == 2);
  if (overlap_mode == 1) {
    HIPCHK(hipEventRecord(c->ev_fork, c->stream));
    HIPCHK(hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
    int rc = chol_of_P(c, c->stream2);
    if (rc) return rc;
    HIPCHK(hipEventRecord(c->ev_join, c->stream2));
  }
  // K1.  Events on the main stream are kept to a minimum (each one costs microseconds between dependent kernels): with
  // the kernel timer on, ev_k0 / ev_k1 bracket K1 and ev_k1 doubles as the fork point; otherwise one untimed fork event.
  if (c->ktimer) HIPCHK(hipEventRecord(c->ev_k0, c->stream));
  c->use_kept_factor = false;
  c->point_nl = 0;
  c->point_boost_n = 0;
  if (overlap_mode == 3 || overlap_mode == 4) {
    ovp::CholJob cj{c->P, c->L, nullptr, nullptr, c->n, c->ld, c->flags, 0, nullptr, 0, 0.0};
    if (c->have_factor && c->Lkeep) {  // the plane loop left M with M M^T = P: no chol(P) (cj.n = 0), the update runs on M
      cj.n = 0;
      c->use_kept_factor = true;
    } else {
      // The batch's information matrix lives on the clones and the estimated calibration.  When everything in front of the first
      // of those columns (the IMU block, dt) is untouched, the factor of P is taken in reversed index order: T = I + L^T A L is
      // then the identity outside its leading n - s0 columns, and both products, chol(T) and the substitution shrink with it
      // (config 2: 194 of 210 - 13 tile columns instead of 14) without a permutation of P.
      const bool no_flip = getenv("OVP_POINT_NO_FLIP") != nullptr;  // (read per call: the tests switch it)
      int s0 = c->n;
      for (int cid : c->h_clone_id) s0 = cid < s0 ? cid : s0;
      if (fp.calmask & 0x3Fu) s0 = c->calib_id < s0 ? c->calib_id : s0;
      if (fp.calmask & (0xFFu << 6)) s0 = c->intr_id < s0 ? c->intr_id : s0;
      for (int dc : c->dense_cols) s0 = dc < s0 ? dc : s0;  // (pending dense blocks may involve columns of their own: a second camera)
      if (!no_flip && s0 >= 8 && s0 < c->n) {
        cj.flip = 1;
        c->point_nl = c->n - s0;
        // the columns in front of the batch's take a relative diagonal boost inside the factorization that the end of the update
        // takes off again (CholJob::boost): exact, and an exact stochastic clone (IMU pose == newest clone) factors on this path
        const bool no_boost = getenv("OVP_POINT_NO_BOOST") != nullptr;  // (read per call: the tests switch it)
        if (!no_boost && s0 <= 64) {
          if (!c->boost) HIPCHK(dalloc(&c->boost, 64));
          cj.boost = c->boost;
          cj.boost_n = s0;
          cj.boost_rel = 1e-9;
          c->point_boost_n = s0;
        }
      }
    }
    if (overlap_mode == 4 && cj.n > 0) {
      ovp::Chol2Job j;
      memset(&j, 0, sizeof(j));
      j.A = c->P;
      j.n = c->n;
      j.ld = c->ld;
      j.mode = 0;
      j.flag = c->flags;
      j.Ldense = c->L;
      j.ldo = c->ld;
      j.flip = cj.flip;
      j.boost = cj.boost;
      j.boost_n = cj.boost_n;
      j.boost_rel = cj.boost_rel;
      HIPCHK(hipEventRecord(c->ev_fork, c->stream));
      c->need_join = true;  // (from here on the side stream may hold work: ovp_build_gate_gram_tail joins on a failing exit)
      HIPCHK(hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
      HIPCHK(ovp_launch_chol2(&j, nullptr, nullptr, c->stream2));
      HIPCHK(hipEventRecord(c->ev_join, c->stream2));
      cj.n = 0;
    }
    HIPCHK(ovp_launch_feat_chol(&fp, &cj, c->stream));
  } else {
    HIPCHK(ovp_launch_feat_gate(&fp, c->stream));
  }
  hipEvent_t fork_ev = c->ev_fork;
  if (c->ktimer) {
    fork_ev = c->ev_k1;
    c->kpending = true;
  }
  // K2 runs on the side stream in mode 2 (it is the shorter of the two branches: the join below then never stalls the
  // main stream, and the cross-queue wake-up latency sits at the START of the side branch, off the critical path)
  hipStream_t s2k = c->stream;
  if (overlap_mode == 2) {
    HIPCHK(hipEventRecord(fork_ev, c->stream));
    HIPCHK(hipStreamWaitEvent(c->stream2, fork_ev, 0));
    s2k = c->stream2;
    int rc = chol_of_P(c, c->stream);
    if (rc) return rc;
  } else {
    if (c->ktimer) HIPCHK(hipEventRecord(c->ev_k1, c->stream));
    if (overlap_mode == 0) {
      int rc = chol_of_P(c, c->stream);
      if (rc) return rc;
      HIPCHK(hipEventRecord(c->ev_join, c->stream));
    }
  }
  // K2 (on the Fa features that own rows of rec / G)
  const int used_chunks = Fa > 0 ? (2 * Fa + c->rows_per_chunk - 1) / c->rows_per_chunk : 0;
  if (Fa > 0) {
    int nsplit = (3 * Fa + 511) / 512;
    if (nsplit < 1) nsplit = 1;
    if (nsplit > c->n_split) nsplit = c->n_split;
    HIPCHK(ovp_launch_gram_pair(c->rec, fp.n_clones, Fa, c->rows_per_chunk, used_chunks, c->gramS, c->G, 3 * Fa, c->ldg,
                                n + 1, c->n_split, c->part, &nsplit, s2k));
    HIPCHK(ovp_launch_reduce_gram(c->gramS, fp.n_clones, used_chunks, c->gramR, s2k));
    HIPCHK(ovp_launch_assemble(c->gramR, fp.n_clones, 1, c->part, nsplit, c->colmap, n, c->Ab, c->ld, s2k));
  } else {
    HIPCHK(hipMemsetAsync(c->Ab, 0, sizeof(double) * (size_t)(n + 1) * c->ld, s2k));
  }
  if (!c->dense_cols.empty()) {
    // the pair of the dense blocks accepted by ovp_msckf_dense_blocks joins the batch's (same update, update/UpdaterMSCKF.cpp:767-814)
    const int m = (int)c->dense_cols.size();
    if (!c->Acc) HIPCHK(dalloc(&c->Acc, (size_t)c->n_max * c->n_max));
    if (!c->bcc) HIPCHK(dalloc(&c->bcc, (size_t)c->n_max));
    HIPCHK(hipMemcpyAsync(c->Acc, c->dense_A.data(), sizeof(double) * (size_t)m * m, hipMemcpyHostToDevice, s2k));
    HIPCHK(hipMemcpyAsync(c->bcc, c->dense_b.data(), sizeof(double) * m, hipMemcpyHostToDevice, s2k));
    HIPCHK(hipMemcpyAsync(c->idbuf, c->dense_cols.data(), sizeof(int) * m, hipMemcpyHostToDevice, s2k));
    HIPCHK(ovp_launch_scatter_gram_add(c->Acc, c->bcc, m, c->idbuf, c->Ab, c->ld, n, s2k));
    HIPCHK(hipStreamSynchronize(s2k));  // (the host vectors are pageable and about to be cleared; slow path)
    c->dense_cols.clear();
  }
  if (overlap_mode == 2) {
    // the Gram pair must be complete on the main stream when this call returns (the caller may all-reduce it there)
    HIPCHK(hipEventRecord(c->ev_join, c->stream2));
    HIPCHK(hipStreamWaitEvent(c->stream, c->ev_join, 0));
  }
  return 0;
}

// Dense blocks beside the batch (see ovplane_hip.h): gates on the host against the marginal of the involved columns, pair kept for
// the point update that follows.
extern "C" int ovp_msckf_dense_blocks(ovp_ctx* c, double chi2_multiplier, int n_blocks, const int* rows, const int* cols, const double* H,
                                      const int* col_ids, const double* res, uint8_t* accepted, double* chi2) {
  if (!c || n_blocks < 0 || (n_blocks > 0 && (!rows || !cols || !H || !col_ids || !res))) return OVP_E_ARG;
  if (!c->have_cov) return OVP_E_STATE;
  c->dense_cols.clear();
  if (n_blocks == 0) return 0;
  // union of the columns, in ascending order
  std::vector<int> uni;
  {
    size_t oc = 0;
    for (int k = 0; k < n_blocks; ++k) {
      if (rows[k] < 1 || cols[k] < 1) return OVP_E_ARG;
      for (int j = 0; j < cols[k]; ++j) {
        const int id = col_ids[oc + j];
        if (id < 0 || id >= c->n) return OVP_E_ARG;
        uni.push_back(id);
      }
      oc += cols[k];
    }
    std::sort(uni.begin(), uni.end());
    uni.erase(std::unique(uni.begin(), uni.end()), uni.end());
  }
  const int m = (int)uni.size();
  std::vector<int> ones((size_t)m, 1), pos((size_t)c->n, -1);
  for (int i = 0; i < m; ++i) pos[uni[i]] = i;
  std::vector<double> Pm((size_t)m * m);
  {
    const int rm = ovp_cov_marginal(c, uni.data(), ones.data(), m, Pm.data());
    if (rm) return rm;
  }
  std::vector<double> A((size_t)m * m, 0.0), b((size_t)m, 0.0);
  size_t oH = 0, oc = 0, orr = 0;
  int n_acc = 0;
  for (int k = 0; k < n_blocks; ++k) {
    const int r = rows[k], q = cols[k];
    const double* Hk = H + oH;        // column-major r x q
    const int* idk = col_ids + oc;
    const double* rk = res + orr;
    oH += (size_t)r * q;
    oc += q;
    orr += r;
    // S = H Pm H^T + I (lower triangle), chi2 = |L^-1 r|^2
    std::vector<double> HP((size_t)r * q, 0.0), S((size_t)r * r, 0.0), y((size_t)r);
    for (int j = 0; j < q; ++j)
      for (int l = 0; l < q; ++l) {
        const double p = Pm[(size_t)pos[idk[l]] * m + pos[idk[j]]];
        if (p == 0.0) continue;
        for (int i = 0; i < r; ++i) HP[(size_t)j * r + i] += Hk[(size_t)l * r + i] * p;
      }
    for (int a = 0; a < r; ++a)
      for (int i = a; i < r; ++i) {
        double v = (i == a) ? 1.0 : 0.0;
        for (int j = 0; j < q; ++j) v += HP[(size_t)j * r + i] * Hk[(size_t)j * r + a];
        S[(size_t)a * r + i] = v;
      }
    double x2 = 0.0;
    bool spd = true;
    for (int j = 0; j < r && spd; ++j) {
      double d = S[(size_t)j * r + j];
      for (int l = 0; l < j; ++l) d -= S[(size_t)l * r + j] * S[(size_t)l * r + j];
      if (!(d > 0.0)) {
        spd = false;
        break;
      }
      d = sqrt(d);
      S[(size_t)j * r + j] = d;
      double rj = rk[j];
      for (int l = 0; l < j; ++l) rj -= S[(size_t)l * r + j] * y[l];
      y[j] = rj / d;
      x2 += y[j] * y[j];
      for (int i = j + 1; i < r; ++i) {
        double v = S[(size_t)j * r + i];
        for (int l = 0; l < j; ++l) v -= S[(size_t)l * r + i] * S[(size_t)l * r + j];
        S[(size_t)j * r + i] = v / d;
      }
    }
    if (!spd) x2 = INFINITY;
    const bool ok = x2 <= chi2_multiplier * ovp_chi2_quantile_095(r);
    if (accepted) accepted[k] = ok ? 1 : 0;
    if (chi2) chi2[k] = x2;
    if (!ok) continue;
    ++n_acc;
    for (int j = 0; j < q; ++j) {
      const int pj = pos[idk[j]];
      double bj = 0.0;
      for (int i = 0; i < r; ++i) bj += Hk[(size_t)j * r + i] * rk[i];
      b[pj] += bj;
      for (int l = 0; l < q; ++l) {
        double v = 0.0;
        for (int i = 0; i < r; ++i) v += Hk[(size_t)j * r + i] * Hk[(size_t)l * r + i];
        A[(size_t)pos[idk[l]] * m + pj] += v;
      }
    }
  }
  if (n_acc > 0) {
    c->dense_cols = uni;
    c->dense_A = A;
    c->dense_b = b;
  }
  return 0;
}

extern "C" int ovp_gram_buffer(ovp_ctx* c, double** Ab_dev, int* n_rows, int* ld) {
  if (!c || !Ab_dev) return OVP_E_ARG;
  *Ab_dev = c->Ab;
  if (n_rows) *n_rows = c->n + 1;
  if (ld) *ld = c->ld;
  return 0;
}

extern "C" int ovp_ekf_update_from_gram_async(ovp_ctx* c) {
  if (!c) return OVP_E_ARG;
  if (!c->have_cov) return OVP_E_STATE;
  int rc = ekf_from_gram(c, true, true);
  if (rc) return rc;
  if (c->ktimer) {
    HIPCHK(hipEventRecord(c->ev_t[3], c->stream));
    c->timed = true;
  }
  return 0;
}

extern "C" int ovp_msckf_fetch_results(ovp_ctx* c, double* dx_host, uint8_t* accepted_host, double* chi2_host,
                                       ovp_update_info* info) {
  if (!c) return OVP_E_ARG;
  const int n = c->n, F = c->n_feats;
  {
    // [flags | dx] are published by the last block of the dx kernel (or, on the fallback path, by a publish kernel);
    // chi2 / accept were written into the pinned block by K1 itself
    unsigned seq = c->pub_seq;
    if (!c->pub_pending) {
      const int words = (int)((16 + sizeof(double) * (size_t)n + 7) / 8);
      seq = ++c->seq;
      hipLaunchKernelGGL(k_publish_results, dim3(1), dim3(1024), 0, c->stream, (unsigned long long*)c->res_block,
                         (unsigned long long*)c->h_res_block_dev, words, (volatile unsigned*)((char*)c->h_res_block_dev +
                         ((char*)c->h_seq - (char*)c->h_res_block)), seq);
      HIPCHK(hipGetLastError());
    }
    c->pub_pending = false;
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (__atomic_load_n((const unsigned*)c->h_seq, __ATOMIC_ACQUIRE) != seq) {
      if ((++spins & 0xFFFu) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {
        HIPCHK(hipStreamSynchronize(c->stream));  // error path: surface a fault instead of spinning forever
        if (__atomic_load_n((const unsigned*)c->h_seq, __ATOMIC_ACQUIRE) != seq) return OVP_E_STATE;
        break;
      }
      __builtin_ia32_pause();
    }
  }
  if (c->h_flags[0]) {
    // chol(P) failed: the prior is only positive semi-definite.  Same update in the reference's S-form (no factor of P needed).
    int rs = ekf_sform(c);
    if (rs) return rs;
  }
  if (dx_host) memcpy(dx_host, c->h_dx, sizeof(double) * n);
  if (accepted_host && F) memcpy(accepted_host, c->h_accept, (size_t)F);
  if (chi2_host && F) memcpy(chi2_host, c->h_chi2, sizeof(double) * F);
  if (c->timed) {
    hipEventSynchronize(c->ev_t[3]);  // recorded behind the publishing kernel: may trail the sequence word by a moment
    // stage times while the kernel timer is on: [0] K1, [1] unused (K2 runs beside chol(P)), [2] chol(P) || K2 and the EKF
    // update, [3] total from the start of K1
    hipEventElapsedTime(&c->last_ms[0], c->ev_k0, c->ev_k1);
    c->last_ms[1] = 0.f;
    hipEventElapsedTime(&c->last_ms[2], c->ev_k1, c->ev_t[3]);
    hipEventElapsedTime(&c->last_ms[3], c->ev_k0, c->ev_t[3]);
    c->timed = false;
  }
  if (c->kpending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, c->ev_k0, c->ev_k1) == hipSuccess) {
      c->ktime_ms += ms;
      c->klaunches += 1;
    }
    c->kpending = false;
  }
  if (info) {
    memset(info, 0, sizeof(*info));
    // n_meas may live in caller-owned device memory (bind_device): read it back once per batch for the row count
    if (F && !c->h_nmeas_valid) {
      c->h_nmeas.resize(F);
      HIPCHK(hipMemcpy(c->h_nmeas.data(), c->fp.n_meas, sizeof(int) * F, hipMemcpyDeviceToHost));
      c->h_nmeas_valid = true;
    }
    for (int f = 0; f < F; ++f)
      if (c->h_accept[f]) {
        info->n_accepted++;
        info->n_rows += 2 * c->h_nmeas[f] - 3;
      }
    info->n_cols = 0;
    info->not_spd = c->h_flags[0];
    info->neg_diag = c->h_flags[1];
  }
  if (c->h_flags[0]) return OVP_E_NOTSPD;
  if (c->h_flags[1]) return OVP_E_NEGDIAG;
  return 0;
}

extern "C" int ovp_msckf_update(ovp_ctx* c, const ovp_update_opts* o, double* dx_host, uint8_t* accepted_host,
                                double* chi2_host, ovp_update_info* info) {
  const double t0 = host_now_ms();
  int rc = ovp_msckf_build_gate_gram_async(c, o);
  if (rc) return rc;
  rc = ovp_ekf_update_from_gram_async(c);
  if (rc) return rc;
  const double t1 = host_now_ms();
  rc = ovp_msckf_fetch_results(c, dx_host, accepted_host, chi2_host, info);
  c->host_acc[4] += t1 - t0;
  c->host_acc[5] += host_now_ms() - t1;
  c->host_acc[6] += 1.0;
  return rc;
}

// ---- StateHelper::EKFUpdate with a dense host H ------------------------------------------------
extern "C" int ovp_ekf_update(ovp_ctx* c, const double* H_host, int rows, int cols, int ld, const int* col_ids,
                              const double* res_host, double* dx_host, ovp_update_info* info) {
  drop_kept_factor(c);  // (writes the covariance: a kept factor no longer belongs to it)
  if (!c || !H_host || !col_ids || !res_host || rows < 1 || cols < 1 || ld < rows) return OVP_E_ARG;
  if (!c->have_cov) return OVP_E_STATE;
  if (cols > c->n) return OVP_E_ARG;
  const int n = c->n;
  for (int j = 0; j < cols; ++j)
    if (col_ids[j] < 0 || col_ids[j] >= n) return OVP_E_ARG;
  // few rows (a frame's landmark re-observations, a zero-velocity update): the reference's own S-form on the kernels of
  // csrc/k_init.hip - S = H P H^T + I in LDS, P+ = P - W W^T - instead of two N x N factorizations
  const char* form_env = getenv("OVP_EKF_INFO_FORM");  // read per call: the tests run both forms in one process
  const bool info_form_only = form_env && form_env[0] == '1';
  if (!info_form_only && rows <= ovp_init_max_rows() && ovp_init_core_lds(0, rows, cols) <= ovp_init_max_lds() && cols <= c->n_max) {
    hipStream_t s = c->stream;
    const size_t oHt = 0, oRes = oHt + (size_t)cols * rows, oId = oRes + rows + 8;
    const size_t bytes = oId * sizeof(double) + sizeof(int) * (size_t)cols + 64;
    const size_t res_doubles = 4 + (size_t)c->n_max + 8;
    int rc = plane2_buffers(c, 0, bytes, res_doubles * sizeof(double));
    if (rc) return rc;
    double* h = (double*)c->pl_hstage;
    double* d = (double*)c->pl_dstage;
    for (int a = 0; a < cols; ++a) memcpy(h + oHt + (size_t)a * rows, H_host + (size_t)a * ld, sizeof(double) * rows);  // = H^T row-major
    memcpy(h + oRes, res_host, sizeof(double) * rows);
    memcpy(h + oId, col_ids, sizeof(int) * cols);
    const int* did = (const int*)(d + oId);
    double* dres = c->smallbuf;
    double* dM = dres + res_doubles;
    double* dLi = dM + (size_t)n * rows;
    double* dy = dLi + (size_t)rows * rows;
    if ((size_t)(dy + rows + 8 - c->smallbuf) > c->small_cap) return OVP_E_CAPACITY;
    HIPCHK(hipMemcpyAsync(c->pl_dstage, c->pl_hstage, bytes, hipMemcpyHostToDevice, s));
    HIPCHK(ovp_launch_init_m(c->P, c->ld, n, did, cols, d + oHt, rows, dM, s));
    HIPCHK(ovp_launch_init_core(c->P, c->ld, n, did, cols, d + oHt, 0, rows, dM, d + oRes /* unused: k = 0 */, d + oRes, d + oRes, 1.0,
                                1e300, dLi, dy, dres, s));
    HIPCHK(ovp_launch_init_update(c->P, c->P_tmp, c->ld, n, dM, rows, 0, rows, dLi, dy, dres, dres + 4, s));
    double* hres = (double*)c->pl_hres;
    {
      const int rf = ovp_fetch_to_hres(c, dres, sizeof(double) * (4 + (size_t)n), s);
      if (rf) return rf;
    }
    if (info) {
      memset(info, 0, sizeof(*info));
      info->n_rows = rows;
      info->n_cols = cols;
      info->not_spd = hres[1] > 0.5 ? 0 : 1;
      info->neg_diag = hres[2] != 0.0;
    }
    if (!(hres[1] > 0.5)) return OVP_E_NOTSPD;  // S = H P H^T + I lost definiteness: P is not a covariance; nothing was written
    double* t = c->P;
    c->P = c->P_tmp;
    c->P_tmp = t;
    if (dx_host) memcpy(dx_host, hres + 4, sizeof(double) * n);
    return hres[2] != 0.0 ? OVP_E_NEGDIAG : 0;
  }
  const size_t need = (size_t)ld * cols;
  if (need > c->Hd_cap) {
    if (c->Hd) hipFree(c->Hd);
    HIPCHK(dalloc(&c->Hd, need));
    c->Hd_cap = need;
  }
  if ((size_t)rows > c->res_cap) {
    if (c->resd) hipFree(c->resd);
    HIPCHK(dalloc(&c->resd, (size_t)rows + 64));
    c->res_cap = (size_t)rows + 64;
  }
  if (!c->Acc) HIPCHK(dalloc(&c->Acc, (size_t)c->n_max * c->n_max));
  if (!c->bcc) HIPCHK(dalloc(&c->bcc, (size_t)c->n_max));
  HIPCHK(hipMemsetAsync(c->flags, 0, sizeof(int) * 4, c->stream));
  HIPCHK(hipMemcpyAsync(c->Hd, H_host, sizeof(double) * need, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(c->resd, res_host, sizeof(double) * rows, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(c->idbuf, col_ids, sizeof(int) * cols, hipMemcpyHostToDevice, c->stream));
  // column-major H [rows x cols, ld] == row-major H^T [cols x rows, ld]:  A = H^T H, b = H^T r
  HIPCHK(ovp_launch_gemm(0, 1, cols, cols, rows, c->Hd, ld, c->Hd, ld, c->Acc, cols, 0, c->stream));
  HIPCHK(ovp_launch_gemm(0, 0, cols, 1, rows, c->Hd, ld, c->resd, 1, c->bcc, 1, 0, c->stream));
  HIPCHK(hipMemsetAsync(c->Ab, 0, sizeof(double) * (size_t)(n + 1) * c->ld, c->stream));
  HIPCHK(ovp_launch_scatter_gram(c->Acc, c->bcc, cols, c->idbuf, c->Ab, c->ld, n, c->stream));
  {
    std::vector<int> ids(col_ids, col_ids + cols);
    std::sort(ids.begin(), ids.end());
    ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
    int rs = set_substate(c, ids);
    if (rs) return rs;
  }
  int rc = ekf_from_gram(c, false);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(c->h_dx, c->dx, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync(c->h_flags, c->flags, sizeof(int) * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (c->h_flags[0]) {  // positive semi-definite prior: S-form instead of the factor of P
    int rs = ekf_sform(c);
    if (rs) return rs;
  }
  if (dx_host) memcpy(dx_host, c->h_dx, sizeof(double) * n);
  if (info) {
    memset(info, 0, sizeof(*info));
    info->n_rows = rows;
    info->n_cols = cols;
    info->not_spd = c->h_flags[0];
    info->neg_diag = c->h_flags[1];
  }
  if (c->h_flags[0]) return OVP_E_NOTSPD;
  if (c->h_flags[1]) return OVP_E_NEGDIAG;
  return 0;
}

