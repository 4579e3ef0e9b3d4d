"""world_size-2 gloo tests of the feature-sharded multi-GPU data path.

What runs here is the PRODUCT's host logic - ov_plane_amd.dist.sharded_update and sharded_plane_then_point_update, the functions
bench.py and a multi-GPU caller use - on two gloo ranks.  Kernels cannot run on a CPU, so each rank drives a stand-in context
(CpuContext below, test infrastructure built on the oracle) that offers the staged interface of capi.Context: shard upload, plane
loop, build + gate + information pair, reduce buffer, update from the summed pair, result fetch.  The checks: every rank ends with
the same covariance, and it is the covariance (and correction) of the unsharded reference flow."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from ov_plane_amd.dist import shard_bounds, sharded_plane_then_point_update, sharded_update  # noqa: E402


def _gram_np(sc, feats, accepted):
    """Information pair of the accepted features of a shard (numpy restatement; accepted is indexed like feats)."""
    from oracle import np_ref as R

    N = sc.N
    A = np.zeros((N, N))
    b = np.zeros(N)
    for k, f in enumerate(feats):
        if not accepted[k]:
            continue
        H_f, H_x, res, order = R.feature_jacobian_full(sc, int(f))
        cols = R.order_cols(order)
        Q1, _ = np.linalg.qr(H_f)
        G, g = Q1.T @ H_x, Q1.T @ res
        A[np.ix_(cols, cols)] += H_x.T @ H_x - G.T @ G
        b[cols] += H_x.T @ res - G.T @ g
    return A, b


class CpuContext:
    """Stand-in for capi.Context on a machine without a GPU: same staged interface, arithmetic by the oracle."""

    def __init__(self, sc):
        from ov_plane_amd.synth import Scene

        self.sc = Scene(sc)
        for k in ("P", "clone_q", "clone_p", "calib_q", "calib_p", "intr", "cp"):
            self.sc[k] = np.array(sc[k], dtype=np.float64, copy=True)
        self.feats = np.arange(sc.F)
        self.n_feats = sc.F
        self._Ab = None

    def batch_upload_scene(self, sc, feats=None):
        self.feats = np.arange(sc.F) if feats is None else np.asarray(feats, dtype=np.int64)
        self.n_feats = len(self.feats)
        self._range = None
        self._used = None
        self.uploads = getattr(self, "uploads", 0) + 1

    def batch_set_range(self, lo=-1, hi=-1):
        self._range = None if (lo == -1 and hi == -1) else (int(lo), int(hi))

    def plane_update(self, opts, plane_of_feat, cp, cp_fej, plane_state_id):
        from oracle import pyoracle

        ref = pyoracle.msckf_plane_update(self.sc)
        for k in ("P", "clone_q", "clone_p", "calib_q", "calib_p", "intr", "cp"):
            self.sc[k] = ref[k]
        self._used = ref["used"]
        return dict(ok=ref["plane_ok"], used=ref["used"], chi2=ref["plane_chi2"], dof=ref["plane_rows"])

    def build_gate_gram_async(self, opts):
        from oracle import pyoracle

        N = self.sc.N
        # what the device walks: the uploaded batch, inside the rank's index range, minus the features of accepted planes
        take = np.ones(self.n_feats, dtype=bool)
        if self._range is not None:
            take &= (np.arange(self.n_feats) >= self._range[0]) & (np.arange(self.n_feats) < self._range[1])
        if getattr(opts, "skip_plane_used", 0):
            assert self._used is not None
            take &= ~self._used[self.feats]
        self._acc, self._chi2 = np.zeros(self.n_feats, dtype=bool), np.zeros(self.n_feats)
        if take.any():
            sel = self.feats[take]
            g = pyoracle.msckf_point_update(self.sc, feats=sel)  # the gate only reads the prior: decisions are per feature
            self._acc[take], self._chi2[take] = g["accepted"], g["chi2"]
            A, b = _gram_np(self.sc, sel, g["accepted"])
        else:
            A, b = np.zeros((N, N)), np.zeros(N)
        self._Ab = torch.from_numpy(np.concatenate([A.ravel(), b]))

    def gram_tensor(self):
        return self._Ab

    def ekf_update_from_gram_async(self):
        N = self.sc.N
        Ab = self._Ab.numpy()
        A, b = Ab[: N * N].reshape(N, N), Ab[N * N:]
        L = np.linalg.cholesky(self.sc.P)
        T = np.eye(N) + L.T @ A @ L
        Y = np.linalg.solve(np.linalg.cholesky(T), L.T).T
        self.sc["P"] = Y @ Y.T
        self._dx = self.sc["P"] @ b

    def fetch_results(self):
        return dict(dx=self._dx, accepted=self._acc, chi2=self._chi2)


def _reference_flow(sc):
    """Unsharded: plane loop, then the point update on the features no accepted plane consumed (oracle, reference loop order)."""
    from oracle import pyoracle
    from ov_plane_amd.synth import Scene

    if sc.cp.shape[0]:
        pl = pyoracle.msckf_plane_update(sc)
        sc2 = Scene(sc)
        for k in ("P", "clone_q", "clone_p", "calib_q", "calib_p", "intr", "cp"):
            sc2[k] = pl[k]
        rest = np.where(~pl["used"])[0]
    else:
        sc2, rest = sc, np.arange(sc.F)
    pt = pyoracle.msckf_point_update(sc2, feats=rest)
    return dict(P=pt["P"], dx=pt["dx"], rest=rest, accepted=pt["accepted"])


def _scene(planes):
    from ov_plane_amd.synth import make_scene

    if planes:
        return make_scene(C=8, F=70, seed=5, n_planes=3, feats_per_plane=12, chi2_mult=99999.0)
    return make_scene(C=7, F=30, seed=3, chi2_mult=1.0)


def _worker(rank, world, port, planes, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ov_plane_amd import capi

        sc = _scene(planes)
        ctx = CpuContext(sc)
        opts = capi.UpdateOpts(1.0, sc.opts["chi2_mult"], sc.opts["sigma_c"], 1, 1, 1, 0)
        if planes:
            pl, pt, mine = sharded_plane_then_point_update(
                ctx, opts, lambda idx: ctx.batch_upload_scene(sc, idx), sc.F, (sc.plane_id, sc.cp, sc.cp_fej, sc.plane_state_id),
                rank=rank, world=world)
            used = pl["used"]
            assert ctx.uploads == 1                      # the frame crosses the bus once per step
            assert not pt["accepted"][used].any()        # point results are indexed like the frame
        else:
            lo, hi = shard_bounds(sc.F, rank, world)
            mine = np.arange(lo, hi)
            ctx.batch_upload_scene(sc, mine)
            pt = sharded_update(ctx, opts)
            used = np.zeros(sc.F, dtype=bool)
        q.put((rank, ctx.sc["P"].copy(), pt["dx"].copy(), mine, pt["accepted"].copy(), used))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _run(planes):
    world = 2
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = 29500 + (os.getpid() % 1000) + (17 if planes else 0)
    procs = [mpc.Process(target=_worker, args=(r, world, port, planes, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_shard_bounds_cover_everything():
    for F in [0, 1, 7, 2000, 8001]:
        for w in [1, 2, 3, 8]:
            b = [shard_bounds(F, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == F
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def test_sharded_update_on_two_gloo_ranks_equals_the_unsharded_update():
    """dist.sharded_update: shards of a 30-feature batch on two ranks, one all-reduce of [A | b], identical update everywhere."""
    res = _run(planes=False)
    ref = _reference_flow(_scene(False))
    shards = np.concatenate([r[3] for r in res])
    assert (np.sort(shards) == np.arange(30)).all()
    acc = np.concatenate([r[4] for r in res])
    assert (acc == ref["accepted"]).all()
    assert np.array_equal(res[0][1], res[1][1])                      # replicas of P are bit-identical
    for _, P, dx, _, _, _ in res:
        assert np.abs(dx - ref["dx"]).max() < 1e-9 and np.abs(P - ref["P"]).max() < 1e-10


def test_plane_loop_replicated_points_sharded_on_two_gloo_ranks():
    """dist.sharded_plane_then_point_update (the shape of BASELINE config 4): every rank runs the plane loop on the whole frame,
    the features no accepted plane consumed are split over the ranks, one all-reduce, identical update."""
    res = _run(planes=True)
    ref = _reference_flow(_scene(True))
    assert (res[0][5] == res[1][5]).all() and res[0][5].sum() > 0    # same planes accepted on both ranks
    shards = np.concatenate([r[3] for r in res])
    assert (np.sort(shards) == ref["rest"]).all()                    # the leftovers, each on exactly one rank
    assert np.array_equal(res[0][1], res[1][1])
    for _, P, dx, _, _, _ in res:
        assert np.abs(dx - ref["dx"]).max() < 1e-9 and np.abs(P - ref["P"]).max() < 1e-10


def test_library_and_python_split_agree_at_config4_size():
    """The two statements of a rank's share - ovp_shard_range_of_mask (what ovp_msckf_update_sharded gives a rank; C, no device
    needed) and dist.leftover_range (what the torch.distributed path and bench.py's stage pass use) - are the same index ranges
    at BASELINE config 4's size (8000 features, 2500 of them on 50 planes) for 1, 2, 4 and 8 ranks, and at the edges: nothing
    consumed, everything consumed, fewer leftovers than ranks."""
    from ov_plane_amd import capi
    from ov_plane_amd.dist import leftover_range

    rng = np.random.default_rng(4)
    F = 8000
    masks = []
    for n_acc in (50, 43, 0):        # all 50 planes accepted, the bench frame's typical count, none
        used = np.zeros(F, dtype=bool)
        on_plane = rng.permutation(F)[:2500].reshape(50, 50)
        used[on_plane[:n_acc].ravel()] = True
        masks.append(used)
    masks += [np.ones(F, dtype=bool), np.r_[np.ones(F - 3, dtype=bool), np.zeros(3, dtype=bool)]]
    for used in masks:
        for world in (1, 2, 4, 8):
            prev_hi, covered = 0, 0
            for rank in range(world):
                lo, hi, mine = leftover_range(used, rank, world)
                assert capi.shard_range_of_mask(used, F, rank, world) == (lo, hi)
                if len(mine):
                    assert lo >= prev_hi and (~used[lo:hi]).sum() == len(mine)   # ranges of consecutive ranks do not overlap
                    prev_hi = hi
                covered += len(mine)
            assert covered == int((~used).sum())
    # no plane loop ran: plain balanced ranges
    for world in (1, 3, 8):
        for rank in range(world):
            assert capi.shard_range_of_mask(None, 2000, rank, world) == shard_bounds(2000, rank, world)
