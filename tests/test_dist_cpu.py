"""world_size-2 gloo test of the feature-sharded multi-GPU data path: every rank reduces its
shard to the information pair (A, b), one all-reduce sums the pairs, every rank applies the identical update.
Kernels cannot run on CPU, so the per-shard pair comes from the numpy restatement (test infrastructure); what is
under test is the host logic: sharding, the reduce, and that the sharded pair equals the unsharded one."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from ov_plane_amd.dist import shard_bounds  # noqa: E402


def _gram_np(sc, feats, accepted):
    from oracle import np_ref as R

    N = sc.N
    A = np.zeros((N, N))
    b = np.zeros(N)
    for f in feats:
        if not accepted[f]:
            continue
        H_f, H_x, res, order = R.feature_jacobian_full(sc, f)
        cols = R.order_cols(order)
        Q1, _ = np.linalg.qr(H_f)
        G, g = Q1.T @ H_x, Q1.T @ res
        A[np.ix_(cols, cols)] += H_x.T @ H_x - G.T @ G
        b[cols] += H_x.T @ res - G.T @ g
    return A, b


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import np_ref as R
    from ov_plane_amd.synth import make_scene

    sc = make_scene(C=7, F=30, seed=3, chi2_mult=1.0)
    full = R.msckf_point_update(sc, use_qr=True)
    lo, hi = shard_bounds(sc.F, rank, world)
    A, b = _gram_np(sc, range(lo, hi), full["accepted"])
    t = torch.from_numpy(np.concatenate([A.ravel(), b]))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    Ab = t.numpy()
    A = Ab[: sc.N * sc.N].reshape(sc.N, sc.N)
    b = Ab[sc.N * sc.N:]
    # replicated update in information form
    L = np.linalg.cholesky(sc.P)
    T = np.eye(sc.N) + L.T @ A @ L
    Y = np.linalg.solve(np.linalg.cholesky(T), L.T).T
    Pn = Y @ Y.T
    dx = Pn @ b
    q.put((rank, float(np.abs(dx - full["dx"]).max()), float(np.abs(Pn - full["P"]).max())))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds_cover_everything():
    for F in [0, 1, 7, 2000, 8001]:
        for w in [1, 2, 3, 8]:
            b = [shard_bounds(F, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == F
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def test_sharded_information_pair_equals_unsharded_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, edx, eP in res:
        assert edx < 1e-9 and eP < 1e-10, (rank, edx, eP)
