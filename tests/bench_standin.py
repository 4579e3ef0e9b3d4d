"""Stand-in backend for bench.py's rank code path on a machine without a GPU (TEST INFRASTRUCTURE; selected only by
`bench.py --standin tests.bench_standin`, driven by tests/test_bench_rank_path_cpu.py).

What it replaces: the device (CPU tensors, gloo instead of RCCL), the HIP context (the oracle-backed CpuContext of
tests/test_dist_cpu.py - kernels cannot run on a CPU) and the communicator of the C entry (an object whose all-reduce is
torch.distributed's).  What it does NOT replace, and what the test is about: bench.py's own main() - the launcher that re-executes
under torch.distributed.run, the process group, the communicator id drawn on rank 0 and carried to the others, the collective
agreement that every rank got one, settle()'s collective decision to go on, the timed windows between barriers, the max over
ranks, the stage pass, the point-only side run and the single JSON line of rank 0.  The split of the leftovers is the LIBRARY's
(ovp_shard_range_of_mask, callable without a device), as in ovp_msckf_update_sharded.

Failure injection (environment, read per call): OVP_STANDIN_FAIL = "uid" (rank 0 cannot draw the id), "comm:<rank>" (that rank's
ncclCommInitRank raises), "pre:<rank>" (that rank finds out beforehand that it cannot enter it): bench.py must end on every rank with the torch.distributed collective instead of hanging."""
from __future__ import annotations

import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SMALL = {
    "config2": dict(C=6, F=24, seed=3, chi2_mult=1.0),
    "config3": dict(C=6, F=40, seed=1, n_planes=2, feats_per_plane=10, chi2_mult=99999.0),
    "config4": dict(C=6, F=48, seed=2, n_planes=3, feats_per_plane=8, chi2_mult=99999.0),
    "points8000": dict(C=6, F=24, seed=3, chi2_mult=1.0),
}


class _Comm:
    """What ovp_msckf_update_sharded needs from an ncclComm_t: a sum all-reduce over the ranks."""

    def __init__(self, uid, rank, world):
        assert isinstance(uid, bytes) and len(uid) == 128
        self.uid, self.rank, self.world = uid, rank, world

    def allreduce(self, t):
        import torch.distributed as dist

        dist.all_reduce(t, op=dist.ReduceOp.SUM)


class _Ctx:
    """The instrumentation calls main() makes on a runner's context (all zeros here: nothing is measured)."""

    def __init__(self, inner):
        self.inner = inner

    def host_timing(self, reset=False):
        return dict(plane_pre_ms=0.0, plane_enqueue_ms=0.0, plane_wait_ms=0.0, plane_loop_device_ms=0.0, point_enqueue_ms=0.0,
                    point_wait_ms=0.0, plane_calls=0, point_calls=0)

    def kernel_timer(self, enable=False, reset=False):
        return 0.0, 0

    def plane_kernel_timer(self, enable=0, reset=False):
        return 0.0, 0

    def timings_ms(self):
        return [0.0, 0.0, 0.0, 0.0]


class Runner:
    """bench.StepRunner's interface over the oracle-backed CpuContext."""

    def __init__(self, capi, sc):
        from test_dist_cpu import CpuContext

        self.capi, self.sc0 = capi, sc
        self.has_planes = sc.cp.shape[0] > 0
        self.opts = capi.UpdateOpts(sc.opts["sigma_px"], sc.opts["chi2_mult"], sc.opts["sigma_c"], 1, 1, 1, 0)
        self.opts_pts = capi.UpdateOpts(sc.opts["sigma_px"], sc.opts["chi2_mult"], sc.opts["sigma_c"], 1, 1, 1, 1 if self.has_planes else 0)
        self._mk = lambda: CpuContext(sc)
        self.cpu = self._mk()
        self.ctx = _Ctx(self.cpu)
        self.shard_size = 0
        self._pl = self._pt = None

    def _fresh(self):
        self.cpu = self._mk()  # every step starts from the same prior
        self.cpu.batch_upload_scene(self.sc0)
        return self.cpu, self.sc0

    def step(self):
        c, sc = self._fresh()
        self._pl = c.plane_update(self.opts, sc.plane_id, sc.cp, sc.cp_fej, sc.plane_state_id) if self.has_planes else None
        c.build_gate_gram_async(self.opts_pts)
        c.ekf_update_from_gram_async()
        self._pt = c.fetch_results()
        return self

    def step_sharded_native(self, comm, rank, world, timing=None):
        """The sequence of ovp_msckf_update_sharded: replicated plane loop, the library's split of the leftovers, build, ONE
        all-reduce of the pair on the communicator, identical update."""
        c, sc = self._fresh()
        t0 = time.perf_counter()
        used = None
        if self.has_planes:
            self._pl = c.plane_update(self.opts, sc.plane_id, sc.cp, sc.cp_fej, sc.plane_state_id)
            used = self._pl["used"]
        t1 = time.perf_counter()
        lo, hi = self.capi.shard_range_of_mask(used, sc.F, rank, world)
        c.batch_set_range(lo, hi)
        c.build_gate_gram_async(self.opts_pts)
        t2 = time.perf_counter()
        if comm is not None:
            comm.allreduce(c.gram_tensor())
        t3 = time.perf_counter()
        c.ekf_update_from_gram_async()
        self._pt = dict(c.fetch_results(), shard=(lo, hi))
        c.batch_set_range(-1, -1)
        t4 = time.perf_counter()
        if timing is not None:
            for k, v in (("plane_loop_ms", t1 - t0), ("points_build_ms", t2 - t1), ("allreduce_ms", t3 - t2), ("update_ms", t4 - t3)):
                if k != "plane_loop_ms" or self.has_planes:
                    timing[k] = timing.get(k, 0.0) + 1e3 * v
        self.shard_size = int((~used[lo:hi]).sum()) if used is not None else hi - lo
        self.last_P, self.last_dx = c.sc["P"].copy(), self._pt["dx"].copy()
        return self

    def step_sharded(self, rank, world, timing=None):
        """The torch.distributed path: the product's own ov_plane_amd.dist functions on this context."""
        from ov_plane_amd.dist import shard_bounds, sharded_plane_then_point_update, sharded_update

        c, sc = self._fresh()
        if self.has_planes:
            pl, pt, mine = sharded_plane_then_point_update(c, self.opts, lambda idx: c.batch_upload_scene(sc, idx), sc.F,
                                                           (sc.plane_id, sc.cp, sc.cp_fej, sc.plane_state_id), rank=rank,
                                                           world=world, timing=timing)
            self.shard_size = len(mine)
        else:
            lo, hi = shard_bounds(sc.F, rank, world)
            c.batch_upload_scene(sc, np.arange(lo, hi))
            self.shard_size = hi - lo
            pl, pt = None, sharded_update(c, self.opts, timing=timing)
            full = np.zeros(sc.F, dtype=bool)
            full[lo:hi] = pt["accepted"]
            pt = dict(pt, accepted=full)
        self.last_P, self.last_dx = c.sc["P"].copy(), pt["dx"].copy()
        return pl, pt

    def results(self):
        return self._pl, self._pt

    def close(self):
        dump = os.environ.get("OVP_STANDIN_DUMP")
        if dump and hasattr(self, "last_P") and self.has_planes == (os.environ.get("OVP_STANDIN_DUMP_PLANES", "1") == "1"):
            np.savez(dump % int(os.environ.get("RANK", "0")), P=self.last_P, dx=self.last_dx)


class Backend:
    name = "standin-cpu"
    dist_backend = "gloo"
    data = "standin (no GPU: bench.py's rank code path under test, nothing measured)"
    device = "cpu"

    def __init__(self, torch, bench):
        self.torch = torch
        from ov_plane_amd import capi

        self.capi = capi
        # a CPU step is milliseconds of oracle: keep the untimed settling phase short
        bench.PREWARM_MIN_STEPS, bench.PREWARM_BLOCK, bench.PREWARM_MAX_S = 4, 2, 2.0

    def available(self):
        return True

    def set_device(self, local_rank):
        pass

    def synchronize(self):
        pass

    def stream_ctx(self, run):
        import contextlib

        return contextlib.nullcontext()

    def make_workload(self, name):
        from ov_plane_amd.synth import make_scene

        kw = dict(SMALL[name])
        if kw.get("n_planes"):
            kw["planes_in_state_frac"] = 0.5
        return make_scene(**kw)

    def make_runner(self, sc, local_rank):
        return Runner(self.capi, sc)

    def unique_id(self):
        if os.environ.get("OVP_STANDIN_FAIL", "") == "uid":
            raise RuntimeError("injected: ncclGetUniqueId failed")
        return os.urandom(128)

    def comm_preflight(self, local_rank):
        if os.environ.get("OVP_STANDIN_FAIL", "") == "pre:%s" % os.environ.get("RANK", "0"):
            raise RuntimeError("injected: RCCL not loadable on this rank")

    def comm_create(self, uid, rank, world, local_rank):
        if os.environ.get("OVP_STANDIN_FAIL", "") == "comm:%d" % rank:
            raise RuntimeError("injected: ncclCommInitRank failed on rank %d" % rank)
        return _Comm(uid, rank, world)

    def comm_destroy(self, comm):
        pass


def make_backend(torch, bench):
    from oracle import pyoracle

    pyoracle.build()
    return Backend(torch, bench)
