// C-ABI shim, part 3 (see ovp_ctx.h): the feature-sharded point update over RCCL.
#include "ovp_ctx.h"

// ---- feature-sharded point update over RCCL ---------------------------------------------------------------------
// One process per GPU; every rank holds the same covariance, pose tables and frame; a rank builds the information pair of its
// share of the point features, ONE ncclAllReduce(sum, f64) of [A | b] on the context's stream puts the summed pair on every rank,
// and every rank applies the identical update to its replica (no broadcast of P+).  RCCL is bound at first use with dlopen: a
// process that already carries an RCCL (torch ships one) shares it instead of loading a second copy, and the library itself keeps
// no link-time dependency on it.
namespace {
struct RcclApi {
  void* h = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, ovp_rccl_id, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool tried = false;
};
RcclApi g_rccl;
const int kNcclFloat64 = 8, kNcclUint8 = 1, kNcclSum = 0;  // ncclDataType_t / ncclRedOp_t of rccl.h (ncclFloat64 = ncclDouble = 8, ncclUint8 = 1, ncclSum = 0)

bool rccl_load() {
  if (g_rccl.tried) return g_rccl.AllReduce != nullptr;
  g_rccl.tried = true;
  const char* names[] = {getenv("OVP_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* nm : names) {
    if (!nm || !*nm) continue;
    void* h = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
    if (!h) continue;
    g_rccl.h = h;
    g_rccl.GetUniqueId = (int (*)(void*))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (int (*)(void**, int, ovp_rccl_id, int))dlsym(h, "ncclCommInitRank");
    g_rccl.CommDestroy = (int (*)(void*))dlsym(h, "ncclCommDestroy");
    g_rccl.AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(h, "ncclAllReduce");
    g_rccl.GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
    if (g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.AllReduce) return true;
    g_rccl = RcclApi();
    g_rccl.tried = true;
  }
  return false;
}
int rccl_rc(int r, const char* what) {
  if (r == 0) return 0;
  fprintf(stderr, "ovplane_hip: %s failed: %s\n", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
  return OVP_E_RCCL;
}
}  // namespace

extern "C" int ovp_rccl_unique_id(ovp_rccl_id* id) {
  if (!id) return OVP_E_ARG;
  if (!rccl_load()) return OVP_E_RCCL;
  return rccl_rc(g_rccl.GetUniqueId(id), "ncclGetUniqueId");
}

extern "C" int ovp_rccl_comm_create(const ovp_rccl_id* id, int rank, int world, int device, void** comm) {
  if (!id || !comm || world < 1 || rank < 0 || rank >= world) return OVP_E_ARG;
  if (!rccl_load()) return OVP_E_RCCL;
  HIPCHK(hipSetDevice(device));
  *comm = nullptr;
  return rccl_rc(g_rccl.CommInitRank(comm, world, *id, rank), "ncclCommInitRank");
}

extern "C" int ovp_rccl_comm_destroy(void* comm) {
  if (!comm) return OVP_E_ARG;
  if (!rccl_load()) return OVP_E_RCCL;
  return rccl_rc(g_rccl.CommDestroy(comm), "ncclCommDestroy");
}

// the collective alone, on the context's stream: for callers that drive the staged entry points themselves
extern "C" int ovp_rccl_allreduce_gram(ovp_ctx* c, void* nccl_comm) {
  if (!c || !nccl_comm) return OVP_E_ARG;
  if (!c->have_cov) return OVP_E_STATE;
  if (!rccl_load()) return OVP_E_RCCL;
  return rccl_rc(g_rccl.AllReduce(c->Ab, c->Ab, (size_t)(c->n + 1) * c->ld, kNcclFloat64, kNcclSum, nccl_comm, c->stream), "ncclAllReduce");
}

// this rank's balanced share of the features the update is about (the ones no accepted plane consumed when a mask is given):
// an index range of the resident batch - consecutive ranks tile it, consumed features inside are masked on the device.  Pure
// arithmetic (no context, no device): the CPU tests hold it against ov_plane_amd/dist.py: leftover_range at BASELINE config 4's size.
extern "C" int ovp_shard_range_of_mask(const uint8_t* used, int n_feats, int rank, int world, int* shard_lo, int* shard_hi) {
  if (!shard_lo || !shard_hi || n_feats < 0 || world < 1 || rank < 0 || rank >= world) return OVP_E_ARG;
  int nr = n_feats;
  if (used) {
    nr = 0;
    for (int f = 0; f < n_feats; ++f) nr += used[f] ? 0 : 1;
  }
  const int base = nr / world, rem = nr % world;
  const int a = rank * base + (rank < rem ? rank : rem), b = a + base + (rank < rem ? 1 : 0);
  *shard_lo = *shard_hi = 0;
  if (b <= a) return 0;
  if (!used) {
    *shard_lo = a;
    *shard_hi = b;
    return 0;
  }
  int k = 0;
  for (int f = 0; f < n_feats; ++f) {
    if (used[f]) continue;
    if (k == a) *shard_lo = f;
    if (k == b - 1) {
      *shard_hi = f + 1;
      break;
    }
    ++k;
  }
  return 0;
}

extern "C" int ovp_shard_range(ovp_ctx* c, const ovp_update_opts* o, int rank, int world, int* shard_lo, int* shard_hi) {
  if (!c || !o || !shard_lo || !shard_hi || world < 1 || rank < 0 || rank >= world) return OVP_E_ARG;
  if (!c->have_batch) return OVP_E_STATE;
  const int F = c->n_feats;
  const bool masked = o->skip_plane_used && c->pl_used_valid && (int)c->h_pl_used.size() == F;
  if (o->skip_plane_used && !masked) return OVP_E_STATE;  // no plane update ran on this batch
  return ovp_shard_range_of_mask(masked ? c->h_pl_used.data() : nullptr, F, rank, world, shard_lo, shard_hi);
}

// Errors of the sharded update are COLLECTIVE: a rank whose build failed still enters the all-reduce (with a zero pair) so that
// its peers do not wait for it forever, and says so in one more f64 word summed behind the pair; every rank then returns an error
// (its own, or OVP_E_PEER) from the same call.  k_peer_flag hands the summed word to the results block (flags[3]).
__global__ void k_peer_flag(const double* __restrict__ word, int* __restrict__ flag3) {
  if (*word != 0.0) *flag3 = 1;
}

extern "C" int ovp_msckf_update_sharded(ovp_ctx* c, const ovp_update_opts* o, void* nccl_comm, int rank, int world, double* dx_host,
                                        uint8_t* accepted_host, double* chi2_host, ovp_update_info* info, int* shard_lo,
                                        int* shard_hi) {
  // argument / state checks: identical on every rank of a correctly driven job (same frame, same options), taken before any
  // collective - a job whose ranks disagree HERE is mis-launched, not failing
  if (!c || !o || world < 1 || rank < 0 || rank >= world) return OVP_E_ARG;
  if (world > 1 && !nccl_comm) return OVP_E_ARG;
  if (!c->have_batch || !c->have_cov) return OVP_E_STATE;
  if (nccl_comm && !rccl_load()) return OVP_E_RCCL;
  int lo = 0, hi = 0;
  {
    const int rs = ovp_shard_range(c, o, rank, world, &lo, &hi);
    if (rs) return rs;
  }
  if (shard_lo) *shard_lo = lo;
  if (shard_hi) *shard_hi = hi;
  const double t0 = host_now_ms();
  const size_t pair_elems = (size_t)(c->n + 1) * c->ld;
  int rc = ovp_batch_set_range(c, lo, hi);
  if (rank != 0 && !c->dense_cols.empty()) {
    // a pending pair of dense blocks (ovp_msckf_dense_blocks: gated identically on every replica) is summed ONCE: rank 0 brings
    // it, the others keep its columns (the same leading block of T everywhere) and contribute zeros
    std::fill(c->dense_A.begin(), c->dense_A.end(), 0.0);
    std::fill(c->dense_b.begin(), c->dense_b.end(), 0.0);
  }
  if (!rc) rc = ovp_msckf_build_gate_gram_async(c, o);
  int rc_coll = 0;
  if (nccl_comm) {
    if (rc) {
      // rank-local failure (a HIP error in the build): a zero pair and a raised word, so that the peers' collective completes
      hipMemsetAsync(c->Ab, 0, sizeof(double) * pair_elems, c->stream);
      static const double one = 1.0;  // (static: the asynchronous copy may read it after this block is left)
      hipMemcpyAsync(c->Ab + pair_elems, &one, sizeof(double), hipMemcpyHostToDevice, c->stream);
    } else {
      hipMemsetAsync(c->Ab + pair_elems, 0, sizeof(double), c->stream);
    }
    rc_coll = rccl_rc(g_rccl.AllReduce(c->Ab, c->Ab, pair_elems + 1, kNcclFloat64, kNcclSum, nccl_comm, c->stream), "ncclAllReduce");
    if (!rc && !rc_coll) {
      hipLaunchKernelGGL(k_peer_flag, dim3(1), dim3(1), 0, c->stream, (const double*)(c->Ab + pair_elems), c->flags + 3);
      rc = (int)hipGetLastError();
    }
  }
  if (!rc) rc = rc_coll;
  if (!rc) rc = ovp_ekf_update_from_gram_async(c);
  const double t1 = host_now_ms();
  if (!rc) {
    rc = ovp_msckf_fetch_results(c, dx_host, accepted_host, chi2_host, info);
    if (c->h_flags[3]) {
      // a peer's build failed: this rank's update ran on the pair of the healthy ranks only - every rank reports the failure and
      // the state is not to be used (the reference treats every failure on this path as fatal, state/StateHelper.cpp:185-187)
      c->have_cov = false;
      rc = OVP_E_PEER;
    }
  }
  c->host_acc[4] += t1 - t0;
  c->host_acc[5] += host_now_ms() - t1;
  c->host_acc[6] += 1.0;
  c->range_lo = c->range_hi = -1;
  return rc;
}

// Per-feature decisions of a sharded update on every rank: ovp_msckf_update_sharded fills accepted / chi2 for its own share only
// (zero elsewhere); a caller that erases rejected features from its feature vector on every replica (the Updater surface:
// update/UpdaterMSCKF.cpp:755-757) completes both arrays here - the shares are disjoint, so a sum is a gather.  Collective.
extern "C" int ovp_rccl_gather_decisions(ovp_ctx* c, void* nccl_comm, uint8_t* accepted_host, double* chi2_host) {
  if (!c || !accepted_host) return OVP_E_ARG;
  if (!c->have_batch) return OVP_E_STATE;
  const int F = c->n_feats;
  if (!nccl_comm || F == 0) return 0;
  if (!rccl_load()) return OVP_E_RCCL;
  // staged through the device result block (its chi2 / accept regions are unused while K1 writes straight to the pinned block)
  HIPCHK(hipMemcpyAsync(c->accept, accepted_host, (size_t)F, hipMemcpyHostToDevice, c->stream));
  if (chi2_host) HIPCHK(hipMemcpyAsync(c->chi2, chi2_host, sizeof(double) * F, hipMemcpyHostToDevice, c->stream));
  int rc = rccl_rc(g_rccl.AllReduce(c->accept, c->accept, (size_t)F, kNcclUint8, kNcclSum, nccl_comm, c->stream), "ncclAllReduce(accept)");
  if (!rc && chi2_host)
    rc = rccl_rc(g_rccl.AllReduce(c->chi2, c->chi2, (size_t)F, kNcclFloat64, kNcclSum, nccl_comm, c->stream), "ncclAllReduce(chi2)");
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(accepted_host, c->accept, (size_t)F, hipMemcpyDeviceToHost, c->stream));
  if (chi2_host) HIPCHK(hipMemcpyAsync(chi2_host, c->chi2, sizeof(double) * F, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

