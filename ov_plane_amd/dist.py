"""Host logic of the feature-sharded multi-GPU update.

Point features are independent until compression (update/UpdaterMSCKF.cpp:695-786), so the batch is split
contiguously over ranks; each rank reduces its shard to the information pair [A | b] on its own GPU, ONE all-reduce
(RCCL over xGMI, (N+1) x ld f64 = ~0.36 MB at N = 210) sums the pairs, and every rank applies the identical EKF
update to its replica of P - no broadcast of P+ is needed.  torch.distributed is plumbing only; the C-ABI stays
torch-free (the reduce buffer is exposed through __cuda_array_interface__)."""
from __future__ import annotations


def shard_bounds(n_feats: int, rank: int, world: int):
    """Contiguous, balanced [lo, hi) of the feature batch owned by `rank`."""
    base, rem = divmod(int(n_feats), int(world))
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


class DeviceBufferView:
    """Zero-copy __cuda_array_interface__ view of a device buffer owned by libovplane_hip.so."""

    def __init__(self, ptr, n_elems, typestr="<f8"):
        self.__cuda_array_interface__ = dict(shape=(int(n_elems),), typestr=typestr, data=(int(ptr), False), version=2,
                                             strides=None)


def _gram_tensor(ctx):
    """torch view of the context's reduce buffer [A | b]: the device buffer of libovplane_hip.so, or whatever tensor a stand-in
    context (the CPU one of tests/test_dist_cpu.py) hands out through gram_tensor()."""
    import torch

    if hasattr(ctx, "gram_tensor"):
        return ctx.gram_tensor()
    ptr, rows, ld = ctx.gram_buffer()
    return torch.as_tensor(DeviceBufferView(ptr, rows * ld), device="cuda")


def _collective_stream(ctx):
    """Context manager that makes the stream the context orders its work on the current torch stream (no-op for a CPU stand-in)."""
    import contextlib

    import torch

    if hasattr(ctx, "stream_handle") and torch.cuda.is_available():
        return torch.cuda.stream(torch.cuda.ExternalStream(ctx.stream_handle()))
    return contextlib.nullcontext()


def sharded_update(ctx, opts, group=None, timing=None):
    """One update step on a rank that already holds its shard (ctx.batch_*), the shared pose tables and P.

    The all-reduce is enqueued on the stream the context orders its work on (ovp_ctx_stream), wrapped as a
    torch.cuda.ExternalStream, so that it runs between the two halves of the staged update without a host sync.
    timing: optional dict; when given, the stages are separated by host synchronisations and their wall times (ms) are added to
    "points_build_ms", "allreduce_ms", "update_ms" (diagnostic passes only - it serialises what normally overlaps).
    Returns ctx.fetch_results()."""
    import torch.distributed as dist

    t = _StageClock(ctx, timing)
    ctx.build_gate_gram_async(opts)
    t.lap("points_build_ms")
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        with _collective_stream(ctx):
            dist.all_reduce(_gram_tensor(ctx), op=dist.ReduceOp.SUM, group=group)
    t.lap("allreduce_ms")
    ctx.ekf_update_from_gram_async()
    out = ctx.fetch_results()
    t.lap("update_ms")
    return out


class _StageClock:
    def __init__(self, ctx, timing):
        import time

        self.ctx, self.timing, self.time = ctx, timing, time
        self.t0 = time.perf_counter() if timing is not None else 0.0

    def lap(self, key):
        if self.timing is None:
            return
        if hasattr(self.ctx, "sync"):
            self.ctx.sync()
        t1 = self.time.perf_counter()
        self.timing[key] = self.timing.get(key, 0.0) + 1e3 * (t1 - self.t0)
        self.t0 = t1


def leftover_range(used, rank, world):
    """Index range [lo, hi) of the whole batch that holds rank's balanced share of the features no accepted plane consumed, and
    those features.  Ranges of consecutive ranks tile the batch; the consumed features inside a range are masked on the device."""
    import numpy as np

    rest = np.nonzero(~np.asarray(used, dtype=bool))[0]
    lo, hi = shard_bounds(len(rest), rank, world)
    mine = rest[lo:hi]
    if len(mine) == 0:
        return 0, 0, mine
    return int(mine[0]), int(mine[-1]) + 1, mine


def sharded_plane_then_point_update(ctx, opts, upload_feats, n_feats, plane_args, rank=0, world=1, group=None, point_opts=None,
                                    timing=None):
    """BASELINE config 4 on several GPUs: the plane loop is sequential across planes and cheap (a few
    hundred microseconds per plane), so EVERY rank runs it on the whole batch - deterministic kernels, identical replicas of
    P and of the pose tables afterwards, no collective - and only the point features that no accepted plane consumed are
    sharded for the point update (one all-reduce, `sharded_update`).

    ONE upload per step: the frame stays resident; a rank's share of the leftovers is an index range of it
    (ovp_batch_set_range) inside which the features of accepted planes are masked by the device-side mask the plane loop left
    (ovp_update_opts::skip_plane_used).

    upload_feats(None): uploads the frame as the context's batch.
    plane_args: (plane_of_feat [n_feats], cp, cp_fej, plane_state_id) as for Context.plane_update.
    Returns (plane results, point results, indices of this rank's point shard); the point results are indexed like the frame."""
    import copy

    t = _StageClock(ctx, timing)
    upload_feats(None)
    out_pl = ctx.plane_update(opts, *plane_args)
    t.lap("plane_loop_ms")
    lo, hi, mine = leftover_range(out_pl["used"][:n_feats], rank, world)
    ctx.batch_set_range(lo, hi)
    po = copy.copy(point_opts if point_opts is not None else opts)
    po.skip_plane_used = 1
    res = sharded_update(ctx, po, group, timing)
    ctx.batch_set_range(-1, -1)
    return out_pl, res, mine
