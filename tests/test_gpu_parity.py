"""GPU parity tests: the HIP path (through the C-ABI) against the oracle on identical seeded inputs.

Tolerances (BASELINE.json north_star): |d state| <= 1e-6 ; covariance 1e-4 relative, measured as
max_ij |dP_ij| / sqrt(P_ii P_jj) (correlation-normalised, NOTES.md §6).  Observed errors are ~1e-10."""
import importlib.util
import os

import numpy as np
import pytest

from ov_plane_amd.synth import make_scene

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL_DX = 1e-6
TOL_P = 1e-4


def _cases():
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLD, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    return mg.CASES


def relP(Pa, Pb):
    d = np.sqrt(np.abs(np.diag(Pb)))
    return float((np.abs(Pa - Pb) / np.outer(d, d)).max())


def run_gpu(capi, sc, feats=None, n_extra=0):
    F = sc.F if feats is None else len(feats)
    ctx = capi.Context(sc.N + n_extra, sc.C, max(F, 1))
    ctx.cov_upload(sc.P)
    ctx.state_upload(sc)
    ctx.batch_upload_scene(sc, feats)
    out = ctx.msckf_update(capi.opts_from_scene(sc))
    out["P"] = ctx.cov_download()
    out["ctx"] = ctx
    return out


@pytest.mark.parametrize("name", ["sim11", "ragged", "nocalib", "nofej", "gate_all", "c30"])
def test_matches_committed_golden_vectors(hiplib, name):
    sc = make_scene(**_cases()[name])
    g = np.load(os.path.join(GOLD, "msckf_%s.npz" % name))
    out = run_gpu(hiplib, sc)
    assert (out["accepted"] == g["accepted"]).all()
    assert np.abs(out["chi2"] - g["chi2"]).max() <= 1e-8 * max(1.0, np.abs(g["chi2"]).max())
    assert np.abs(out["dx"] - g["dx"]).max() < TOL_DX
    assert relP(out["P"], g["P"]) < TOL_P
    assert out["info"].n_accepted == int(g["accepted"].sum())
    out["ctx"].close()


@pytest.mark.parametrize("kw", [
    dict(C=11, F=150, seed=21, chi2_mult=1.0),
    dict(C=5, F=33, seed=22, ragged=True, min_meas=2, chi2_mult=1.0),     # includes 2-observation features (dof 1)
    dict(C=31, F=64, seed=23, chi2_mult=1.0),                               # max_clones+1 window
    dict(C=12, F=90, seed=24, ragged=True, chi2_mult=0.6),                  # many rejections
    dict(C=10, F=70, seed=25, chi2_mult=1.0, calib=False, do_fej=False),
    dict(C=7, F=2300, seed=26, ragged=True, min_meas=3, chi2_mult=1.0),     # more features than one round of the fused K1 launch (2040)
    dict(C=9, F=700, seed=27, chi2_mult=1.0),                                # several waves per workgroup, not all eight
    dict(C=9, F=120, seed=28, chi2_mult=1.0, fisheye=True),                  # equidistant lens (ext CamEqui)
    dict(C=31, F=40, seed=29, chi2_mult=1.0, fisheye=True),                  # ... on the all-VALU K1 variant
])
def test_matches_oracle_on_fresh_scenes(hiplib, oracle, kw):
    sc = make_scene(**kw)
    ref = oracle.msckf_point_update(sc)
    out = run_gpu(hiplib, sc)
    # gate decisions: identical unless a feature sits within 1e-9 relative of its threshold
    diff = np.where(out["accepted"] != ref["accepted"])[0]
    assert len(diff) == 0, diff
    assert np.abs(out["chi2"] - ref["chi2"]).max() <= 1e-8 * max(1.0, np.abs(ref["chi2"]).max())
    assert np.abs(out["dx"] - ref["dx"]).max() < TOL_DX
    assert relP(out["P"], ref["P"]) < TOL_P
    assert np.abs(out["P"] - out["P"].T).max() == 0.0
    out["ctx"].close()


@pytest.mark.parametrize("pose,intr", [(True, False), (False, True)])
def test_calibration_column_sets_can_be_switched_separately(hiplib, oracle, pose, intr):
    """do_calib_camera_pose / do_calib_camera_intrinsics decide the column set independently (update/UpdaterHelper.cpp:216-227,
    426-440); the variables stay in the state and are still corrected through their correlations."""
    sc = make_scene(C=9, F=60, seed=41, chi2_mult=1.0)
    sc.opts["do_calib_pose"], sc.opts["do_calib_intr"] = pose, intr
    ref = oracle.msckf_point_update(sc)
    out = run_gpu(hiplib, sc)
    assert (out["accepted"] == ref["accepted"]).all()
    assert np.abs(out["chi2"] - ref["chi2"]).max() <= 1e-8 * max(1.0, np.abs(ref["chi2"]).max())
    assert np.abs(out["dx"] - ref["dx"]).max() < TOL_DX
    assert relP(out["P"], ref["P"]) < TOL_P
    out["ctx"].close()


def test_empty_and_all_rejected_batches(hiplib):
    sc = make_scene(C=6, F=12, seed=31, chi2_mult=1e-9)  # everything fails the gate
    out = run_gpu(hiplib, sc)
    assert out["accepted"].sum() == 0
    assert np.abs(out["dx"]).max() < 1e-12
    assert relP(out["P"], sc.P) < 1e-10
    out["ctx"].close()
    # features with < 2 observations are dropped (UpdaterMSCKF.cpp:94-96)
    sc = make_scene(C=6, F=10, seed=32, chi2_mult=1.0)
    sc.n_meas[:] = 1
    out = run_gpu(hiplib, sc)
    assert out["accepted"].sum() == 0 and np.abs(out["dx"]).max() == 0.0
    out["ctx"].close()


def test_full_size_properties(hiplib):
    """BASELINE config[1] size (30 clones x 2000 features): size-independent properties instead of the oracle."""
    sc = make_scene(C=30, F=2000, seed=0, chi2_mult=1.0)
    out = run_gpu(hiplib, sc)
    P1 = out["P"]
    assert out["accepted"].mean() > 0.9
    assert np.abs(P1 - P1.T).max() == 0.0
    w = np.linalg.eigvalsh(P1)
    assert w.min() > 0
    # information never decreases: P - P+ is PSD
    wd = np.linalg.eigvalsh(sc.P - P1)
    assert wd.min() > -1e-9 * np.abs(wd).max()
    # information identity: P+^-1 - P^-1 = A  (A as accumulated on the device)
    ctx = out["ctx"]
    ld = ((sc.N + 15) // 16) * 16
    Ab = ctx.debug_read("Ab", (sc.N + 1, ld))
    A, b = Ab[: sc.N, : sc.N], Ab[sc.N, : sc.N]
    assert np.abs(A - A.T).max() <= 1e-9 * np.abs(A).max()
    lhs = P1 @ (np.linalg.inv(sc.P) + A)
    assert np.abs(lhs - np.eye(sc.N)).max() < 1e-6
    assert np.abs(out["dx"] - P1 @ b).max() < 1e-9
    ctx.close()
    # feature order does not matter (the stacked update is permutation invariant)
    perm = np.random.default_rng(0).permutation(sc.F)
    out2 = run_gpu(hiplib, sc, feats=perm)
    assert (out2["accepted"] == out["accepted"][perm]).all()
    assert np.abs(out2["dx"] - out["dx"]).max() < 1e-9
    assert relP(out2["P"], P1) < 1e-8
    out2["ctx"].close()
    # splitting the batch: information adds, so two half-batch covariance updates == one full update
    ctx = hiplib.Context(sc.N, sc.C, sc.F)
    ctx.cov_upload(sc.P)
    ctx.state_upload(sc)
    o = hiplib.opts_from_scene(sc)
    o.chi2_multiplier = 1e12  # gate against the *prior* would differ in the second half; disable it here
    ctx.batch_upload_scene(sc, np.arange(0, 1000))
    ctx.msckf_update(o)
    ctx.batch_upload_scene(sc, np.arange(1000, 2000))
    ctx.msckf_update(o)
    P_two = ctx.cov_download()
    ctx.cov_upload(sc.P)
    ctx.batch_upload_scene(sc)
    ctx.msckf_update(o)
    P_one = ctx.cov_download()
    assert relP(P_two, P_one) < 1e-7
    ctx.close()


def _whole_step_against_oracle(hiplib, oracle, sc, ens_kw=None):
    """One UpdaterMSCKF::update downstream of triangulation at full size, device against oracle: the plane loop with the gate at
    chi2_multipler = 1 (the device loop runs on the oracle's accept / reject sequence - ovp_plane_batch::force_decision - so both
    hand the same state to what follows; the plane statistic itself is covered by the plane-gate tests), then the point update on
    every feature no accepted plane consumed (device-side mask) against the all-cores oracle (oracle/ovp_oracle_omp.c, pinned
    against the one-thread restatement in test_oracle_pins) at the state the oracle's plane loop left.  Returns the report."""
    from ov_plane_amd.synth import Scene

    has_planes = sc.cp.shape[0] > 0
    ctx = hiplib.Context(sc.N, sc.C, sc.F)
    ctx.cov_upload(sc.P)
    ctx.state_upload(sc)
    ctx.batch_upload_scene(sc)
    o = hiplib.opts_from_scene(sc)
    rep = {}
    if has_planes:
        ref_pl = oracle.msckf_plane_update(sc)
        pl = ctx.plane_update(o, sc.plane_id, sc.cp, sc.cp_fej, sc.plane_state_id, force_decision=ref_pl["plane_ok"].astype(np.uint8))
        assert (pl["ok"] == ref_pl["plane_ok"]).all() and (pl["used"] == ref_pl["used"]).all()
        assert (pl["dof"] == ref_pl["plane_rows"]).all()
        rep["planes_rejected"] = int((~ref_pl["plane_ok"]).sum())
        # the decisions the device's own statistic would have taken: equal to the oracle's except next to the threshold
        thr = np.array([hiplib.lib().ovp_chi2_quantile_095(int(k)) for k in pl["dof"]])
        differ = (pl["chi2"] <= thr) != ref_pl["plane_ok"]
        assert np.abs(pl["chi2"] - ref_pl["plane_chi2"]).max() <= GATE_BAND
        if ens_kw is not None:
            # the frame is in the ensemble fixture: a would-be flip is allowed only where the four builds of the oracle have not
            # decided the plane themselves (test_plane_gate_against_the_oracle_ensemble)
            ens = _ensemble_of(ens_kw)
            assert (ens["ok"] == ref_pl["plane_ok"]).all()
            assert not (differ & ens["decided"]).any(), np.where(differ & ens["decided"])[0]
            rep["planes_decided_by_the_ensemble"] = int(ens["decided"].sum())
        else:
            assert differ.sum() <= max(2, len(thr) // 12) and (np.abs(ref_pl["plane_chi2"] - thr)[differ] < GATE_BAND).all()
        sc2 = Scene(sc)
        for k in ("P", "clone_q", "clone_p", "calib_q", "calib_p", "intr", "cp"):
            sc2[k] = ref_pl[k]
        rest = np.where(~ref_pl["used"])[0]
        o.skip_plane_used = 1
    else:
        sc2, rest = sc, np.arange(sc.F)
    ref = oracle.msckf_point_update_omp(sc2, feats=rest)
    out = ctx.msckf_update(o)
    P = ctx.cov_download()
    ctx.close()
    acc_d = np.asarray(out["accepted"]).astype(bool)
    assert (acc_d[rest] == ref["accepted"]).all(), np.where(acc_d[rest] != ref["accepted"])[0]
    if has_planes:
        assert not acc_d[ref_pl["used"]].any()
    assert np.abs(out["chi2"][rest] - ref["chi2"]).max() <= 1e-7 * max(1.0, np.abs(ref["chi2"]).max())
    rep["dx_err"] = float(np.abs(out["dx"] - ref["dx"]).max())
    rep["P_err"] = relP(P, ref["P"])
    rep["points_gated"] = int(len(rest))
    rep["points_accepted"] = int(ref["accepted"].sum())
    assert rep["dx_err"] < TOL_DX and rep["P_err"] < TOL_P, rep
    return rep


def test_config2_full_step_matches_oracle(hiplib, oracle):
    """BASELINE config[1] at full size (30 clones, 2000 MSCKF point features, gate at multiplier 1) against the oracle: accept set,
    chi2 of every feature, correction and covariance."""
    sc = make_scene(C=30, F=2000, seed=0, chi2_mult=1.0)
    assert sc.N == 210
    rep = _whole_step_against_oracle(hiplib, oracle, sc)
    assert rep["points_gated"] == 2000 and 1900 < rep["points_accepted"] < 2000


def test_config3_whole_step_matches_oracle(hiplib, oracle):
    """BASELINE config[2] at full size - the bench's timed frame: 20 planes x 50 features + 1000 free points, gate at multiplier 1
    on both levels - plane loop AND the point update on the leftovers against the oracle."""
    kw = dict(C=30, F=2000, seed=0, n_planes=20, feats_per_plane=50, planes_in_state_frac=0.5, chi2_mult=1.0)
    sc = make_scene(**kw)
    assert sc.N == 240
    rep = _whole_step_against_oracle(hiplib, oracle, sc, ens_kw=kw)
    assert rep["planes_rejected"] >= 1 and rep["points_gated"] >= 1000 + 50 * rep["planes_rejected"]


def test_config4_whole_step_matches_oracle(hiplib, oracle):
    """BASELINE config[3] on one GPU at full size (8000 features of which 2500 on 50 planes, N = 285), whole step against the oracle."""
    kw = dict(C=30, F=8000, seed=0, n_planes=50, feats_per_plane=50, planes_in_state_frac=0.5, chi2_mult=1.0)
    sc = make_scene(**kw)
    assert sc.N == 285
    rep = _whole_step_against_oracle(hiplib, oracle, sc, ens_kw=kw)
    assert rep["points_gated"] >= 5500


@pytest.mark.parametrize("C", [31, 32])
def test_plane_constraint_of_a_full_length_track(hiplib, oracle, C):
    """On-plane features observed from 31 / 32 clones (62 / 64 bearing rows: the wavefront is full, the merged point-on-plane row
    update/UpdaterHelper.cpp:503-511 is carried as a wave-uniform term of the sums, not in a lane of its own): plane loop
    against the oracle.  update/UpdaterMSCKF.cpp:413-649 has no limit on the track length; until round 4 32 observations were
    OVP_E_CAPACITY."""
    sc = make_scene(C=C, F=120, seed=50 + C, n_planes=3, feats_per_plane=20, planes_in_state_frac=0.67, chi2_mult=99999.0)
    assert int(sc.n_meas.max()) == C
    ref = oracle.msckf_plane_update(sc)
    ctx = hiplib.Context(sc.N, sc.C, sc.F)
    ctx.cov_upload(sc.P)
    ctx.state_upload(sc)
    ctx.batch_upload_scene(sc)
    out = ctx.plane_update(hiplib.opts_from_scene(sc), sc.plane_id, sc.cp, sc.cp_fej, sc.plane_state_id)
    assert out["rc"] == 0 and (out["ok"] == ref["plane_ok"]).all() and out["ok"].all()
    assert (out["used"] == ref["used"]).all() and (out["dof"] == ref["plane_rows"]).all()
    assert np.abs(out["chi2"] - ref["plane_chi2"]).max() < GATE_BAND
    cq, cpos, calq, calp, intr, cp = _apply_plane_dx(sc, out["dx"], out["ok"])
    assert np.abs(cpos - ref["clone_p"]).max() < TOL_DX and np.abs(cq - ref["clone_q"]).max() < TOL_DX
    assert np.abs(intr - ref["intr"]).max() < TOL_DX and np.abs(cp - ref["cp"]).max() < TOL_DX
    assert relP(ctx.cov_download(), ref["P"]) < TOL_P
    ctx.close()


def _frame_through_the_two_updates(hiplib, sc):
    ctx = hiplib.Context(sc.N, sc.C, sc.F)
    ctx.cov_upload(sc.P)
    ctx.state_upload(sc)
    ctx.batch_upload_scene(sc)
    o = hiplib.opts_from_scene(sc)
    pl = ctx.plane_update(o, sc.plane_id, sc.cp, sc.cp_fej, sc.plane_state_id) if sc.cp.shape[0] else None
    o.skip_plane_used = 1 if pl is not None else 0
    pt = ctx.msckf_update(o)
    P = ctx.cov_download()
    ht = ctx.host_timing()
    ctx.close()
    return pl, pt, P, ht


@pytest.mark.parametrize("switch,kw", [
    # plane loop in the state's own column order on all n columns against the loop's order / leading blocks
    ("OVP_PL_NATURAL_ORDER", dict(C=12, F=260, seed=61, n_planes=6, feats_per_plane=25, planes_in_state_frac=0.5, chi2_mult=1.0)),
    # point update behind a plane loop: chol(P) again against the factor the loop left
    ("OVP_NO_KEPT_FACTOR", dict(C=12, F=260, seed=62, n_planes=5, feats_per_plane=30, planes_in_state_frac=0.6, chi2_mult=1.0)),
    # point update: chol(P) in the state's index order (full-size T) against the reversed-order factor (leading block of T)
    ("OVP_POINT_NO_FLIP", dict(C=20, F=400, seed=63, chi2_mult=1.0)),
    ("OVP_POINT_NO_FLIP", dict(C=7, F=90, seed=64, ragged=True, chi2_mult=1.0, n_slam=5)),   # landmarks behind the clones: the block grows
])
def test_round4_shortcuts_change_nothing_but_rounding(hiplib, monkeypatch, switch, kw):
    """The three shortcuts of round 4 - the plane loop on the leading block of its own column order, the point update on the
    factor that loop left, the point update's T on its leading block through a reversed-order factor of P - are algebraic
    identities: each against its switched-off form on the same frame (same decisions, corrections and covariance to rounding)."""
    sc = make_scene(**kw)
    monkeypatch.delenv(switch, raising=False)
    pl1, pt1, P1, ht1 = _frame_through_the_two_updates(hiplib, sc)
    monkeypatch.setenv(switch, "1")
    pl0, pt0, P0, ht0 = _frame_through_the_two_updates(hiplib, sc)
    if pl1 is not None:
        assert (pl1["ok"] == pl0["ok"]).all() and (pl1["used"] == pl0["used"]).all() and pl1["ok"].any()
        assert np.abs(pl1["chi2"] - pl0["chi2"]).max() < 1e-6 * max(1.0, np.abs(pl0["chi2"]).max())
        assert np.abs(pl1["dx"] - pl0["dx"]).max() < 1e-9
    assert (pt1["accepted"] == pt0["accepted"]).all() and pt1["accepted"].sum() > 10
    assert np.abs(pt1["dx"] - pt0["dx"]).max() < 1e-9
    assert relP(P1, P0) < 1e-8
    # the host clocks of the entry points are there and make sense
    assert ht1["point_calls"] == 1 and ht1["point_enqueue_ms"] > 0 and ht1["point_wait_ms"] >= 0
    if pl1 is not None:
        assert ht1["plane_calls"] >= 1 and ht1["plane_enqueue_ms"] >= ht1["plane_pre_ms"] >= 0


def test_config3_plane_loop_at_full_size_matches_oracle(hiplib, oracle):
    """BASELINE config[2]: 30 clones, 2000 features of which 1000 lie on 20 planes (10 of them in the state, N = 240).  The plane
    loop is cheap enough for the oracle at full size (about a second); the point update on the 1000 free points is checked
    through its information identity."""
    sc = make_scene(C=30, F=2000, seed=0, n_planes=20, feats_per_plane=50, planes_in_state_frac=0.5, chi2_mult=99999.0)
    assert sc.N == 240
    ref = oracle.msckf_plane_update(sc)
    ctx = hiplib.Context(sc.N, sc.C, sc.F)
    ctx.cov_upload(sc.P)
    ctx.state_upload(sc)
    ctx.batch_upload_scene(sc)
    o = hiplib.opts_from_scene(sc)
    out = ctx.plane_update(o, sc.plane_id, sc.cp, sc.cp_fej, sc.plane_state_id)
    assert (out["ok"] == ref["plane_ok"]).all() and out["ok"].all()
    assert (out["used"] == ref["used"]).all() and out["used"].sum() == 1000
    cq, cpos, calq, calp, intr, cp = _apply_plane_dx(sc, out["dx"], out["ok"])
    assert np.abs(cpos - ref["clone_p"]).max() < TOL_DX and np.abs(cq - ref["clone_q"]).max() < TOL_DX
    assert np.abs(intr - ref["intr"]).max() < TOL_DX and np.abs(cp - ref["cp"]).max() < TOL_DX
    P_pl = ctx.cov_download()
    assert relP(P_pl, ref["P"]) < TOL_P
    # point update on the free points, at the state the plane loop left on the device
    rest = np.where(~out["used"])[0]
    ctx.batch_upload_scene(sc, rest)
    o.chi2_multiplier = 1.0
    upd = ctx.msckf_update(o)
    P1 = ctx.cov_download()
    assert upd["accepted"].mean() > 0.9
    ld = ((sc.N + 15) // 16) * 16
    Ab = ctx.debug_read("Ab", (sc.N + 1, ld))
    A, b = Ab[: sc.N, : sc.N], Ab[sc.N, : sc.N]
    assert np.abs(P1 @ (np.linalg.inv(P_pl) + A) - np.eye(sc.N)).max() < 1e-6
    assert np.abs(upd["dx"] - P1 @ b).max() < 1e-9
    ctx.close()


def test_config4_on_one_gpu_plane_loop_and_point_update(hiplib, oracle):
    """BASELINE config[3] on a single GPU, where it fits: 30 clones, 8000 features of which 2500 lie on 50 planes (25 of them in the
    state, N = 285 - the largest state of the BASELINE configurations, above 16 tile rows).  Plane loop against the oracle (about
    ten seconds for it), then the point update on the 5500 free points straight from the device-side used mask
    (ovp_update_opts::skip_plane_used), checked through its information identity."""
    sc = make_scene(C=30, F=8000, seed=0, n_planes=50, feats_per_plane=50, planes_in_state_frac=0.5, chi2_mult=99999.0)
    assert sc.N == 285 and int((sc.plane_id > 0).sum()) == 2500
    ref = oracle.msckf_plane_update(sc)
    ctx = hiplib.Context(sc.N, sc.C, sc.F)
    ctx.cov_upload(sc.P)
    ctx.state_upload(sc)
    ctx.batch_upload_scene(sc)
    o = hiplib.opts_from_scene(sc)
    out = ctx.plane_update(o, sc.plane_id, sc.cp, sc.cp_fej, sc.plane_state_id)
    assert (out["ok"] == ref["plane_ok"]).all() and out["ok"].all()
    assert (out["used"] == ref["used"]).all() and out["used"].sum() == 2500
    assert (out["dof"] == ref["plane_rows"]).all()
    cq, cpos, calq, calp, intr, cp = _apply_plane_dx(sc, out["dx"], out["ok"])
    assert np.abs(cpos - ref["clone_p"]).max() < TOL_DX and np.abs(cq - ref["clone_q"]).max() < TOL_DX
    assert np.abs(intr - ref["intr"]).max() < TOL_DX and np.abs(cp - ref["cp"]).max() < TOL_DX
    P_pl = ctx.cov_download()
    assert relP(P_pl, ref["P"]) < TOL_P
    # point update on the free points: same batch, the features of the accepted planes are skipped on the device
    o.chi2_multiplier = 1.0
    o.skip_plane_used = 1
    upd = ctx.msckf_update(o)
    P1 = ctx.cov_download()
    assert not upd["accepted"][out["used"]].any()
    assert upd["accepted"][~out["used"]].mean() > 0.9
    ld = ((sc.N + 15) // 16) * 16
    Ab = ctx.debug_read("Ab", (sc.N + 1, ld))
    A, b = Ab[: sc.N, : sc.N], Ab[sc.N, : sc.N]
    assert np.abs(P1 @ (np.linalg.inv(P_pl) + A) - np.eye(sc.N)).max() < 1e-6
    assert np.abs(upd["dx"] - P1 @ b).max() < 1e-9
    # the same point update with the leftovers uploaded as a batch of their own: identical information pair
    rest = np.where(~out["used"])[0]
    ctx.cov_upload(P_pl)
    ctx.batch_upload_scene(sc, rest)
    o.skip_plane_used = 0
    upd2 = ctx.msckf_update(o)
    assert (upd2["accepted"] == upd["accepted"][rest]).all()
    assert np.abs(upd2["dx"] - upd["dx"]).max() < 1e-9
    ctx.close()


def test_config4_plane_gate_at_multiplier_one(hiplib, oracle):
    """BASELINE config[3]'s plane loop with the gate deciding (chi2_multipler = 1, the value of the real-data configs): the device
    loop runs on the oracle's accept / reject sequence, so both see the same state at every one of the 50 planes (N = 285: the
    two-workgroup solve); state and covariance to the path's tolerances, the device statistic within the rounding-decided band
    of the reference's (profiles/r03_plane_gate_agreement.json: two builds of the oracle differ by up to 18), the decisions it
    would have taken equal to the oracle's except next to the threshold."""
    sc = make_scene(C=30, F=8000, seed=1, n_planes=50, feats_per_plane=50, planes_in_state_frac=0.5, chi2_mult=1.0)
    ref = oracle.msckf_plane_update(sc)
    assert 3 <= (~ref["plane_ok"]).sum() <= 25          # the gate is deciding at this multiplier
    ctx = hiplib.Context(sc.N, sc.C, sc.F)
    ctx.cov_upload(sc.P)
    ctx.state_upload(sc)
    ctx.batch_upload_scene(sc)
    out = ctx.plane_update(hiplib.opts_from_scene(sc), sc.plane_id, sc.cp, sc.cp_fej, sc.plane_state_id,
                           force_decision=ref["plane_ok"].astype(np.uint8))
    assert (out["ok"] == ref["plane_ok"]).all() and (out["used"] == ref["used"]).all() and (out["dof"] == ref["plane_rows"]).all()
    cq, cpos, calq, calp, intr, cp = _apply_plane_dx(sc, out["dx"], out["ok"])
    assert np.abs(cpos - ref["clone_p"]).max() < TOL_DX and np.abs(cq - ref["clone_q"]).max() < TOL_DX
    assert np.abs(intr - ref["intr"]).max() < TOL_DX and np.abs(cp - ref["cp"]).max() < TOL_DX
    assert relP(ctx.cov_download(), ref["P"]) < TOL_P
    d = out["chi2"] - ref["plane_chi2"]
    assert np.abs(d).max() < GATE_BAND and abs(d.mean()) < 2.0, (d.mean(), np.abs(d).max())
    thr = np.array([hiplib.lib().ovp_chi2_quantile_095(int(k)) for k in out["dof"]])
    differ = (out["chi2"] <= thr) != ref["plane_ok"]
    # a would-be flip only where the four builds of the oracle have not decided the plane themselves (ensemble fixture)
    ens = _ensemble_of(dict(C=30, F=8000, seed=1, n_planes=50, feats_per_plane=50, planes_in_state_frac=0.5, chi2_mult=1.0))
    assert (ens["ok"] == ref["plane_ok"]).all() and ens["decided"].sum() >= 40
    assert not (differ & ens["decided"]).any(), np.where(differ & ens["decided"])[0]
    ctx.close()


@pytest.mark.parametrize("case", ["exact_clone", "zero_variance", "stochastic_clone", "stochastic_clone_no_boost"])
def test_positive_semidefinite_priors_are_updated_in_s_form(hiplib, oracle, case, monkeypatch):
    """state/StateHelper.cpp:159-187 never factors P, so the reference updates a covariance that is only positive SEMI-definite:
    right after StateHelper::clone the new pose is an exact copy of the IMU pose (:346-396), and a variable may carry a
    zero-variance prior.  The device's fast path factors P; when that fails it falls back to the S-form on the Cholesky factor of the
    batch's information matrix (ekf_sform) instead of returning OVP_E_NOTSPD.  Both priors below are exactly singular."""
    sc = make_scene(C=9, F=80, seed=41, chi2_mult=1.0)
    P = sc.P.copy()
    if case.startswith("stochastic_clone"):
        # what every frame of a running filter sees: the newest clone is an exact copy of the IMU pose (columns 0..5), which lies
        # in FRONT of the batch's columns.  Round 5: the reversed-order chol(P) boosts the diagonal there and the update's last kernel
        # takes the amounts off again ((P + D)+ = P+ + D for H D = 0), so this prior stays on the fast path; with the boost switched
        # off the S-form retry gives the same answer
        if case.endswith("no_boost"):
            monkeypatch.setenv("OVP_POINT_NO_BOOST", "1")
        b = sc.ids["clones"][-1]
        idx = np.arange(sc.N)
        idx[b:b + 6] = np.arange(0, 6)
        P = P[np.ix_(idx, idx)]
    elif case == "exact_clone":
        a, b = sc.ids["clones"][-2], sc.ids["clones"][-1]  # the newest clone becomes an exact copy of the one before it
        idx = np.arange(sc.N)
        idx[b:b + 6] = np.arange(a, a + 6)
        P = P[np.ix_(idx, idx)]
        sc["clone_q"][-1], sc["clone_p"][-1] = sc["clone_q"][-2], sc["clone_p"][-2]
        sc["clone_q_fej"][-1], sc["clone_p_fej"][-1] = sc["clone_q_fej"][-2], sc["clone_p_fej"][-2]
    else:
        k = sc.ids["intr"] + 4  # first distortion coefficient known exactly
        P[k, :] = 0.0
        P[:, k] = 0.0
    assert np.linalg.eigvalsh(P).min() < 1e-12 * np.linalg.eigvalsh(P).max()
    sc["P"] = P
    ref = oracle.msckf_point_update(sc)
    out = run_gpu(hiplib, sc)
    assert out["rc"] == 0
    assert (out["accepted"] == ref["accepted"]).all()
    assert np.abs(out["dx"] - ref["dx"]).max() < TOL_DX
    d = np.sqrt(np.abs(np.diag(ref["P"])))
    d[d == 0] = 1.0
    assert (np.abs(out["P"] - ref["P"]) / np.outer(d, d)).max() < TOL_P
    out["ctx"].close()


@pytest.mark.parametrize("kw", [
    dict(C=30, F=300, seed=71, ragged=True, min_meas=2, chi2_mult=1.0),    # every track length 2..30: one to four 16-column blocks
    dict(C=30, F=120, seed=72, chi2_mult=0.8),                               # full-length tracks (60 rows + border = 64), rejections
    dict(C=15, F=90, seed=73, ragged=True, min_meas=2, chi2_mult=1.0),     # corner columns that straddle two blocks (n = 14, 30)
    dict(C=8, F=60, seed=74, chi2_mult=1.0, calib=False),                    # no calibration columns
])
def test_k1_with_b_in_lds_and_through_the_scratch_agree(hiplib, oracle, monkeypatch, kw):
    """Round 5: K1 builds B = H_x P H_x^T + I block by block inside the bordered factorization (operand rows and the factor's
    sub-diagonal tiles share the wave's LDS); OVP_K1_BSCR=1 is the form of rounds 1-4 that sends B through a per-feature scratch in
    device memory.  Same decisions, chi2 to 1e-8 of the oracle in both, and the two forms agree with each other to rounding."""
    sc = make_scene(**kw)
    assert len(set(np.asarray(sc.n_meas).tolist())) > (3 if kw.get("ragged") else 0)
    ref = oracle.msckf_point_update(sc)
    monkeypatch.delenv("OVP_K1_BSCR", raising=False)
    new = run_gpu(hiplib, sc)
    monkeypatch.setenv("OVP_K1_BSCR", "1")
    old = run_gpu(hiplib, sc)
    scale = np.maximum(1.0, np.abs(ref["chi2"]))
    for out in (new, old):
        assert (out["accepted"] == ref["accepted"]).all()
        assert (np.abs(out["chi2"] - ref["chi2"]) / scale).max() <= 1e-8
        assert np.abs(out["dx"] - ref["dx"]).max() < TOL_DX
        assert relP(out["P"], ref["P"]) < TOL_P
    assert (np.abs(new["chi2"] - old["chi2"]) / scale).max() <= 1e-10
    assert np.abs(new["dx"] - old["dx"]).max() < 1e-9 and relP(new["P"], old["P"]) < 1e-8
    new["ctx"].close()
    old["ctx"].close()


@pytest.mark.parametrize("case", ["regular", "stochastic_clone", "many_features"])
def test_chol_p_on_the_side_stream_equals_the_fused_launch(hiplib, oracle, monkeypatch, case):
    """Round 5: up to 1976 features chol(P) of the point update is a k_chol2 launch (mode 0 with the reversed order and the diagonal
    boost of CholJob) on the side stream, beside feature workgroups that leave a CU on every XCD (OVP_OVERLAP_MODE=4, default);
    mode 3 keeps the factorization in workgroup 0 of the fused launch.  Same update to rounding - including on the prior every frame
    of a running filter sees (newest clone == IMU pose: exactly singular, factored through the boost) - and the oracle's answer."""
    if case == "many_features":
        sc = make_scene(C=12, F=1900, seed=76, ragged=True, min_meas=3, chi2_mult=1.0)  # a round of 247 workgroups, eight waves each
    else:
        sc = make_scene(C=9, F=80, seed=75, chi2_mult=1.0)
    if case == "stochastic_clone":
        b = sc.ids["clones"][-1]
        idx = np.arange(sc.N)
        idx[b:b + 6] = np.arange(0, 6)
        sc["P"] = sc.P[np.ix_(idx, idx)]
        assert np.linalg.eigvalsh(sc.P).min() < 1e-12 * np.linalg.eigvalsh(sc.P).max()
    ref = oracle.msckf_point_update(sc) if case != "many_features" else None
    outs = {}
    for mode in ("4", "3"):
        monkeypatch.setenv("OVP_OVERLAP_MODE", mode)
        outs[mode] = run_gpu(hiplib, sc)
        assert outs[mode]["rc"] == 0
    a, b_ = outs["4"], outs["3"]
    assert (a["accepted"] == b_["accepted"]).all() and a["accepted"].sum() > 10
    assert np.abs(a["dx"] - b_["dx"]).max() < 1e-9
    d = np.sqrt(np.abs(np.diag(b_["P"])))
    d[d == 0] = 1.0
    assert (np.abs(a["P"] - b_["P"]) / np.outer(d, d)).max() < 1e-8
    assert np.abs(a["P"] - a["P"].T).max() == 0.0
    if ref is not None:
        assert (a["accepted"] == ref["accepted"]).all()
        assert np.abs(a["dx"] - ref["dx"]).max() < TOL_DX
        dr = np.sqrt(np.abs(np.diag(ref["P"])))
        dr[dr == 0] = 1.0
        assert (np.abs(a["P"] - ref["P"]) / np.outer(dr, dr)).max() < TOL_P
    for o in outs.values():
        o["ctx"].close()


@pytest.mark.parametrize("case", ["exact_clone", "zero_variance", "stochastic_clone"])
@pytest.mark.parametrize("big", [False, True])
def test_plane_loop_on_a_positive_semidefinite_prior(hiplib, oracle, case, big):
    """The plane loop on the covariance StateHelper::clone leaves (state/StateHelper.cpp:346-396: the newest pose an EXACT copy,
    P singular) - the reference never factors P (:159-187) and updates it as it is (update/UpdaterMSCKF.cpp:413-649).  The device
    loop factors P0 once; when that fails it runs the same loop on the pivot-dropping factor of the unit-diagonal form
    (P_k = L0 (I + L0^T A L0)^-1 L0^T holds for any L0 L0^T = P0) instead of returning OVP_E_NOTSPD.  `big`: above the
    factorization's limit, where the loop runs on the marginal of the involved columns."""
    kw = dict(C=9, F=150, seed=43, n_planes=3, feats_per_plane=25, planes_in_state_frac=0.67, chi2_mult=99999.0)
    if big:
        kw.update(n_slam=70)  # 70 free landmarks in the state: N = 300
    sc = make_scene(**kw)
    assert (sc.N > 288) == big
    P = sc.P.copy()
    if case == "stochastic_clone":
        # newest clone == IMU pose (columns 0..5, which no plane involves): the loop's diagonal boost on the uninvolved columns keeps
        # this prior - the one a running filter has in every frame - on the first attempt
        b = sc.ids["clones"][-1]
        idx = np.arange(sc.N)
        idx[b:b + 6] = np.arange(0, 6)
        P = P[np.ix_(idx, idx)]
    elif case == "exact_clone":
        a, b = sc.ids["clones"][-2], sc.ids["clones"][-1]
        idx = np.arange(sc.N)
        idx[b:b + 6] = np.arange(a, a + 6)
        P = P[np.ix_(idx, idx)]
        sc["clone_q"][-1], sc["clone_p"][-1] = sc["clone_q"][-2], sc["clone_p"][-2]
        sc["clone_q_fej"][-1], sc["clone_p_fej"][-1] = sc["clone_q_fej"][-2], sc["clone_p_fej"][-2]
    else:
        k = sc.ids["intr"] + 4  # first distortion coefficient known exactly
        P[k, :] = 0.0
        P[:, k] = 0.0
    assert np.linalg.eigvalsh(P).min() < 1e-12 * np.linalg.eigvalsh(P).max()
    sc["P"] = P
    ref = oracle.msckf_plane_update(sc)
    ctx = hiplib.Context(sc.N, sc.C, sc.F)
    ctx.cov_upload(sc.P)
    ctx.state_upload(sc)
    ctx.batch_upload_scene(sc)
    out = ctx.plane_update(hiplib.opts_from_scene(sc), sc.plane_id, sc.cp, sc.cp_fej, sc.plane_state_id)
    assert (out["ok"] == ref["plane_ok"]).all() and out["ok"].all()
    assert (out["used"] == ref["used"]).all()
    cq, cpos, calq, calp, intr, cp = _apply_plane_dx(sc, out["dx"], out["ok"])
    assert np.abs(cpos - ref["clone_p"]).max() < TOL_DX and np.abs(cq - ref["clone_q"]).max() < TOL_DX
    assert np.abs(intr - ref["intr"]).max() < TOL_DX and np.abs(cp - ref["cp"]).max() < TOL_DX
    d = np.sqrt(np.abs(np.diag(ref["P"])))
    d[d == 0] = 1.0
    assert (np.abs(ctx.cov_download() - ref["P"]) / np.outer(d, d)).max() < TOL_P
    # the point update on the rest follows on the same singular covariance (its own S-form fallback)
    o = hiplib.opts_from_scene(sc)
    o.chi2_multiplier = 1.0
    o.skip_plane_used = 1
    upd = ctx.msckf_update(o)
    assert upd["rc"] == 0 and upd["accepted"][~out["used"]].mean() >= 0.5
    ctx.close()


@pytest.mark.parametrize("info_form", ["0", "1"])
def test_dense_ekf_update_matches_reference_form(hiplib, info_form, monkeypatch):
    """ovp_ekf_update == StateHelper::EKFUpdate (state/StateHelper.cpp:121-202) for an arbitrary dense H: the S-form kernels
    that take few-row updates (csrc/k_init.hip) and the information form behind them (OVP_EKF_INFO_FORM=1)."""
    from oracle import np_ref

    monkeypatch.setenv("OVP_EKF_INFO_FORM", info_form)

    rng = np.random.default_rng(4)
    sc = make_scene(C=6, F=4, seed=41)
    order = [(int(sc.ids["clones"][1]), 6), (int(sc.ids["calib"]), 6), (0, 3)]
    cols = np_ref.order_cols(order)
    H = rng.standard_normal((9, len(cols))) * 30.0
    res = rng.standard_normal(9)
    Pn, dx = np_ref.ekf_update(sc.P, order, H, res)
    ctx = hiplib.Context(sc.N, sc.C, 4)
    ctx.cov_upload(sc.P)
    dxg, info = ctx.ekf_update(H, cols, res)
    Pg = ctx.cov_download()
    assert np.abs(dxg - dx).max() < 1e-9
    assert relP(Pg, Pn) < 1e-8
    ctx.close()


@pytest.mark.parametrize("info_form", ["0", "1"])
@pytest.mark.parametrize("case", ["landmark_update", "delayed_init_rows", "everything"])
def test_dense_ekf_update_above_the_tile_limit(hiplib, case, info_form, monkeypatch):
    """(info_form = 0: the few-row S-form kernels of csrc/k_init.hip, which do not depend on N; 1: the information form.)
    N = 366 (30 clones + 52 landmarks) is past the register-resident factorization (N <= 288).  Measurements that touch at
    most 288 columns take the sub-state update (factor P[s,s], then P -= G (A - A Pss+ A) G^T); one that touches every column
    falls back to the global-memory kernels.  Both against the reference form."""
    from oracle import np_ref
    from ov_plane_amd.synth import make_slam_scene

    monkeypatch.setenv("OVP_EKF_INFO_FORM", info_form)
    rng = np.random.default_rng(11)
    sc = make_slam_scene(C=30, n_slam=52, seed=3)
    assert sc.N == 366
    if case == "landmark_update":      # newest clone, calibration, one landmark
        order = [(int(sc.ids["clones"][-1]), 6), (int(sc.ids["calib"]), 6), (int(sc.ids["intr"]), 8), (int(sc.ids["slam"][7]), 3)]
        rows = 2
    elif case == "delayed_init_rows":   # a full track: every clone + calibration
        order = [(int(sc.ids["calib"]), 6), (int(sc.ids["intr"]), 8)] + [(int(c), 6) for c in sc.ids["clones"]]
        rows = 59
    else:
        order = [(0, sc.N)]
        rows = 40
    cols = np_ref.order_cols(order)
    H = rng.standard_normal((rows, len(cols))) * 20.0
    res = rng.standard_normal(rows)
    Pn, dx = np_ref.ekf_update(sc.P, order, H, res)
    ctx = hiplib.Context(sc.N + 8, sc.C + 2, 4)
    ctx.cov_upload(sc.P)
    dxg, info = ctx.ekf_update(H, cols, res)
    Pg = ctx.cov_download()
    # the sub-state form subtracts (A - A Pss+ A, P - G Lambda G^T): a few digits less than the factor form, far inside 1e-4
    assert np.abs(dxg - dx).max() < 1e-7 * max(1.0, np.abs(dx).max())
    assert relP(Pg, Pn) < 1e-6
    assert np.abs(Pg - Pg.T).max() < 1e-12 * np.abs(Pg).max()
    ctx.close()


def test_covariance_bookkeeping(hiplib):
    """propagate / clone / marginalise / marginal-gather against the restatement (StateHelper.cpp:41-119,231-396)."""
    from oracle import np_ref

    rng = np.random.default_rng(6)
    sc = make_scene(C=5, F=4, seed=51)
    N = sc.N
    ctx = hiplib.Context(N + 18, sc.C + 2, 4)
    ctx.cov_upload(sc.P)
    Phi = np.eye(15) + 0.05 * rng.standard_normal((15, 15))
    Qh = rng.standard_normal((15, 15)) * 1e-3
    Q = Qh @ Qh.T
    Pref = np_ref.ekf_propagation(sc.P, 0, 15, [(0, 15)], Phi, Q)
    neg = ctx.cov_propagate(0, [0], [15], Phi, Q)
    assert neg == 0
    Pprop = ctx.cov_download()
    assert np.abs(Pprop - Pref).max() < 1e-12 * max(1.0, np.abs(Pref).max())
    Pref = Pprop  # the copies below must be bit-exact with respect to what is resident on the device
    # clone the IMU pose (StateHelper::clone)
    ctx.cov_clone(0, 6)
    Pc = ctx.cov_download()
    assert Pc.shape == (N + 6, N + 6)
    assert np.abs(Pc[:N, :N] - Pref).max() == 0.0
    assert np.abs(Pc[N:, N:] - Pref[:6, :6]).max() == 0.0   # an exact copy, bit for bit (state/StateHelper.cpp:346-396)
    assert np.abs(Pc[:N, N:] - Pref[:, :6]).max() == 0.0 and np.abs(Pc[N:, :N] - Pref[:6, :]).max() == 0.0
    # the optional inflation of the new block's diagonal (off by default): keeps an exact clone off the singular-prior paths
    ctx.cov_clone_jitter(1e-11)
    ctx.cov_clone(0, 6)
    Pj = ctx.cov_download()
    blk = Pref[:6, :6].copy()
    blk[np.diag_indices(6)] *= (1.0 + 1e-11)
    assert np.abs(Pj[N + 6:, N + 6:] - blk).max() <= 1e-16 * np.abs(blk).max()
    ctx.cov_clone_jitter(0.0)
    ctx.cov_marginalize(N + 6, 6)
    assert np.abs(ctx.cov_download() - Pc).max() == 0.0
    # marginal covariance gather
    ids = [int(sc.ids["clones"][2]), 16]
    M = ctx.cov_marginal(ids, [6, 6])
    cols = np_ref.order_cols([(ids[0], 6), (16, 6)])
    assert np.abs(M - Pc[np.ix_(cols, cols)]).max() == 0.0
    # marginalise the oldest clone
    cid = int(sc.ids["clones"][0])
    ctx.cov_marginalize(cid, 6)
    Pm = ctx.cov_download()
    keep = [i for i in range(N + 6) if not (cid <= i < cid + 6)]
    assert np.abs(Pm - Pc[np.ix_(keep, keep)]).max() == 0.0
    ctx.close()


@pytest.mark.parametrize("n_slam", [10, 25, 40])
def test_larger_states_use_every_factorization_path(hiplib, oracle, n_slam):
    """N = 30 + 6C + 3 n_slam: 240 (15-slot register-resident Cholesky), 285 (25-slot), 330 (global-memory fallback)."""
    sc = make_scene(C=30, F=48, seed=61 + n_slam, n_slam=n_slam, chi2_mult=1.0)
    ref = oracle.msckf_point_update(sc)
    out = run_gpu(hiplib, sc)
    assert (out["accepted"] == ref["accepted"]).all()
    assert np.abs(out["dx"] - ref["dx"]).max() < TOL_DX
    assert relP(out["P"], ref["P"]) < TOL_P
    out["ctx"].close()


def _apply_plane_dx(sc, dxs, oks):
    """Host side of the plane loop: ext Type::update applied in plane order (what the caller of the C-ABI does)."""
    from ov_plane_amd.synth import quat_boxplus

    cq, cpos = sc.clone_q.copy(), sc.clone_p.copy()
    calq, calp, intr, cp = sc.calib_q.copy(), sc.calib_p.copy(), sc.intr.copy(), sc.cp.copy()
    for pl in range(dxs.shape[0]):
        if not oks[pl]:
            continue
        dx = dxs[pl]
        for i in range(sc.C):
            cid = sc.ids["clones"][i]
            cq[i] = quat_boxplus(cq[i], dx[cid:cid + 3])
            cpos[i] = cpos[i] + dx[cid + 3:cid + 6]
        calq = quat_boxplus(calq, dx[16:19])
        calp = calp + dx[19:22]
        intr = intr + dx[22:30]
        for k in range(sc.cp.shape[0]):
            sid = sc.plane_state_id[k]
            if sid >= 0:
                cp[k] = cp[k] + dx[sid:sid + 3]
    return cq, cpos, calq, calp, intr, cp


@pytest.mark.parametrize("kw", [
    dict(C=11, F=160, seed=5, n_planes=4, feats_per_plane=25, chi2_mult=99999.0),
    dict(C=9, F=120, seed=8, n_planes=6, feats_per_plane=12, chi2_mult=99999.0, ragged=True),
    dict(C=30, F=200, seed=9, n_planes=4, feats_per_plane=30, chi2_mult=99999.0),
    dict(C=8, F=100, seed=10, n_planes=4, feats_per_plane=20, chi2_mult=99999.0, fisheye=True),
])
def test_plane_loop_matches_oracle(hiplib, oracle, kw):
    """UpdaterMSCKF.cpp:411-649 (planes in and out of the state) followed by the point update on the leftover features."""
    sc = make_scene(**kw)
    ref = oracle.msckf_plane_update(sc)
    ctx = hiplib.Context(sc.N, sc.C, sc.F)
    ctx.cov_upload(sc.P)
    ctx.state_upload(sc)
    ctx.batch_upload_scene(sc)
    o = hiplib.opts_from_scene(sc)
    out = ctx.plane_update(o, sc.plane_id, sc.cp, sc.cp_fej, sc.plane_state_id)
    assert (out["ok"] == ref["plane_ok"]).all()
    assert (out["used"] == ref["used"]).all()
    assert (out["dof"][ref["plane_rows"] > 0] == ref["plane_rows"][ref["plane_rows"] > 0]).all()
    cq, cpos, calq, calp, intr, cp = _apply_plane_dx(sc, out["dx"], out["ok"])
    assert np.abs(cpos - ref["clone_p"]).max() < TOL_DX
    assert np.abs(cq - ref["clone_q"]).max() < TOL_DX
    assert np.abs(calp - ref["calib_p"]).max() < TOL_DX and np.abs(calq - ref["calib_q"]).max() < TOL_DX
    assert np.abs(intr - ref["intr"]).max() < TOL_DX
    assert np.abs(cp - ref["cp"]).max() < TOL_DX
    P = ctx.cov_download()
    assert relP(P, ref["P"]) < TOL_P
    # the gate statistic: deterministic part + expectation of the reference's rounding-decided rows - within the distance two builds
    # of the oracle keep from each other (test_plane_gate_against_the_oracle_ensemble), on every plane that ran
    ran = ref["plane_rows"] > 0
    assert np.abs(out["chi2"][ran] - ref["plane_chi2"][ran]).max() <= GATE_BAND, (out["chi2"][ran], ref["plane_chi2"][ran])
    ctx.close()


def _oracle_full_update(oracle, sc, slam=None):
    """Reference flow of UpdaterMSCKF::update downstream of triangulation: plane loop, then the point loop on the
    features the planes did not consume, at the state/covariance the plane loop left behind."""
    from ov_plane_amd.synth import Scene

    if sc.cp.shape[0] > 0:
        pl = oracle.msckf_plane_update(sc, slam=slam)
    else:
        pl = dict(P=sc.P, clone_q=sc.clone_q, clone_p=sc.clone_p, calib_q=sc.calib_q, calib_p=sc.calib_p, intr=sc.intr,
                  cp=sc.cp, used=np.zeros(sc.F, dtype=bool))
    sc2 = Scene(sc)
    for k in ("P", "clone_q", "clone_p", "calib_q", "calib_p", "intr", "cp"):
        sc2[k] = pl[k]
    rest = np.where(~pl["used"])[0]
    pt = oracle.msckf_point_update(sc2, feats=rest)
    from ov_plane_amd.synth import quat_boxplus

    dx = pt["dx"]
    cq, cpos = pl["clone_q"].copy(), pl["clone_p"].copy()
    for i in range(sc.C):
        cid = sc.ids["clones"][i]
        cq[i] = quat_boxplus(cq[i], dx[cid:cid + 3])
        cpos[i] = cpos[i] + dx[cid + 3:cid + 6]
    cp = pl["cp"].copy()
    for k in range(cp.shape[0]):
        sid = sc.plane_state_id[k]
        if sid >= 0:
            cp[k] = cp[k] + dx[sid:sid + 3]
    kept = np.zeros(sc.F, dtype=bool)
    kept[rest[pt["accepted"]]] = True
    out = dict(P=pt["P"], clone_q=cq, clone_p=cpos, calib_q=quat_boxplus(pl["calib_q"], dx[16:19]),
               calib_p=pl["calib_p"] + dx[19:22], intr=pl["intr"] + dx[22:30], cp=cp, used=pl["used"], kept=kept)
    if slam is not None and len(slam["id"]):
        out["slam_p"] = pl["slam_p"] + np.array([dx[i:i + 3] for i in slam["id"]])
    return out


@pytest.mark.parametrize("kw,k_rows", [
    (dict(C=8, F=90, seed=73, n_planes=3, feats_per_plane=15, n_slam=3, chi2_mult=99999.0, ragged=True), 3),
    (dict(C=11, F=160, seed=5, n_planes=4, feats_per_plane=25, n_slam=5, chi2_mult=99999.0), 5),
    (dict(C=9, F=100, seed=12, n_planes=4, feats_per_plane=20, n_slam=4, chi2_mult=99999.0, do_fej=False), 4),
    # 22 landmarks on ONE out-of-state plane (round 4 stopped at 16: OVP_PLANE_MAX_SLAM; update/UpdaterMSCKF.cpp:232-252 has no limit)
    (dict(C=9, F=120, seed=14, n_planes=2, feats_per_plane=30, planes_in_state_frac=0.5, n_slam=24, chi2_mult=99999.0), 22),
])
def test_plane_loop_with_slam_landmarks_matches_oracle(hiplib, oracle, kw, k_rows):
    """SLAM landmarks lying on planes that are not in the state take part in that plane's update with one constraint row whose
    feature Jacobian stays in the landmark's columns (update/UpdaterMSCKF.cpp:232-252, :545-552); the landmarks move with
    every accepted plane."""
    from ov_plane_amd.synth import slam_rows_on_planes

    sc = make_scene(**kw)
    slam = slam_rows_on_planes(sc, k_rows)
    ref = oracle.msckf_plane_update(sc, slam=slam)
    base = oracle.msckf_plane_update(sc)
    assert np.abs(ref["P"] - base["P"]).max() > 1e-9  # the rows matter
    ctx = hiplib.Context(sc.N, sc.C, sc.F)
    ctx.cov_upload(sc.P)
    ctx.state_upload(sc)
    ctx.batch_upload_scene(sc)
    out = ctx.plane_update(hiplib.opts_from_scene(sc), sc.plane_id, sc.cp, sc.cp_fej, sc.plane_state_id, slam=slam)
    assert (out["ok"] == ref["plane_ok"]).all() and ref["plane_ok"].all()
    assert (out["used"] == ref["used"]).all()
    assert (out["dof"] == ref["plane_rows"]).all()
    cq, cpos, calq, calp, intr, cp = _apply_plane_dx(sc, out["dx"], out["ok"])
    assert np.abs(cpos - ref["clone_p"]).max() < TOL_DX and np.abs(cq - ref["clone_q"]).max() < TOL_DX
    assert np.abs(intr - ref["intr"]).max() < TOL_DX and np.abs(cp - ref["cp"]).max() < TOL_DX
    lm = slam["p"].copy()
    for k in range(out["dx"].shape[0]):
        if out["ok"][k]:
            for q, i in enumerate(slam["id"]):
                lm[q] += out["dx"][k][i:i + 3]
    assert np.abs(lm - ref["slam_p"]).max() < TOL_DX
    assert relP(ctx.cov_download(), ref["P"]) < TOL_P
    ctx.close()


@pytest.mark.parametrize("kw,k_rows", [
    (dict(C=11, F=200, seed=31, n_planes=16, feats_per_plane=10, planes_in_state_frac=0.75, n_slam=60, chi2_mult=99999.0), 0),
    (dict(C=11, F=200, seed=32, n_planes=16, feats_per_plane=10, planes_in_state_frac=0.75, n_slam=60, chi2_mult=1.0), 6),
    (dict(C=30, F=300, seed=33, n_planes=6, feats_per_plane=40, planes_in_state_frac=0.5, n_slam=27, chi2_mult=1.0), 0),
])
def test_plane_loop_above_the_factorization_limit_runs_on_the_involved_columns(hiplib, oracle, kw, k_rows):
    """update/UpdaterMSCKF.cpp:413-649 has no size limit.  States above the tile factorization (config/sim shape: 11 clones, 60
    landmarks, 12 planes in the state -> N = 312; 30 clones + 27 landmarks -> N = 300) run the plane loop on the columns its planes
    involve and carry the rest along (ovp_api_plane.hip: plane_update_substate): decisions (the oracle's imposed where the gate is
    active), state corrections - also of the variables no plane touches (IMU state, free landmarks) -, covariance of the whole
    state against the oracle."""
    from ov_plane_amd.synth import slam_rows_on_planes

    sc = make_scene(**kw)
    assert sc.N > 288
    slam = slam_rows_on_planes(sc, k_rows) if k_rows else None
    ref = oracle.msckf_plane_update(sc, slam=slam)
    ctx = hiplib.Context(sc.N, sc.C, sc.F)
    ctx.cov_upload(sc.P)
    ctx.state_upload(sc)
    ctx.batch_upload_scene(sc)
    out = ctx.plane_update(hiplib.opts_from_scene(sc), sc.plane_id, sc.cp, sc.cp_fej, sc.plane_state_id, slam=slam,
                           force_decision=ref["plane_ok"].astype(np.uint8))
    assert out["rc"] == 0
    assert (out["ok"] == ref["plane_ok"]).all() and ref["plane_ok"].sum() >= 3
    assert (out["used"] == ref["used"]).all() and (out["dof"] == ref["plane_rows"]).all()
    cq, cpos, calq, calp, intr, cp = _apply_plane_dx(sc, out["dx"], out["ok"])
    assert np.abs(cpos - ref["clone_p"]).max() < TOL_DX and np.abs(cq - ref["clone_q"]).max() < TOL_DX
    assert np.abs(intr - ref["intr"]).max() < TOL_DX and np.abs(cp - ref["cp"]).max() < TOL_DX
    if slam is not None:
        lm = slam["p"].copy()
        for k in range(out["dx"].shape[0]):
            if out["ok"][k]:
                for q, i in enumerate(slam["id"]):
                    lm[q] += out["dx"][k][i:i + 3]
        assert np.abs(lm - ref["slam_p"]).max() < TOL_DX
    P1 = ctx.cov_download()
    assert relP(P1, ref["P"]) < TOL_P
    # the variables outside the loop's columns move through their correlation with them: dx = P0[:, s] P0[s, s]^-1 dx[s]
    inv = np.zeros(sc.N, dtype=bool)
    for i in range(sc.C):
        inv[sc.ids["clones"][i]: sc.ids["clones"][i] + 6] = True
    inv[16:30] = True
    for sid in sc.plane_state_id:
        if sid >= 0:
            inv[sid:sid + 3] = True
    if slam is not None:
        for i in slam["id"]:
            inv[i:i + 3] = True
    assert inv.sum() <= 287 and (~inv).sum() >= 15
    Gs = sc.P[:, inv] @ np.linalg.inv(sc.P[np.ix_(inv, inv)])
    for k in np.where(out["ok"])[0]:
        full = Gs @ out["dx"][k][inv]
        assert np.abs(full - out["dx"][k]).max() < 1e-8 * max(1.0, np.abs(full).max()), k
    assert np.abs(out["dx"][~out["ok"]]).max(initial=0.0) == 0.0
    # and the point update that follows works on the whole state as before
    o = hiplib.opts_from_scene(sc)
    o.chi2_multiplier = 1.0
    o.skip_plane_used = 1
    upd = ctx.msckf_update(o)
    assert not upd["accepted"][out["used"]].any() and upd["accepted"][~out["used"]].mean() > 0.8
    ctx.close()


@pytest.mark.parametrize("kw", [
    dict(C=11, F=120, seed=71, chi2_mult=1.0),                                           # points only
    dict(C=10, F=150, seed=72, n_planes=4, feats_per_plane=20, chi2_mult=99999.0),         # planes + points
    dict(C=8, F=90, seed=73, n_planes=3, feats_per_plane=15, chi2_mult=99999.0, ragged=True),
    dict(C=9, F=80, seed=74, chi2_mult=1.0, fisheye=True),
])
def test_host_cpp_mirror_updater_msckf(hiplib, oracle, kw):
    """ov_plane::UpdaterMSCKF::update / StateHelper (C++ host classes, ov_plane_amd/csrc/host) over the C-ABI vs the oracle,
    including the feature-vector side effects (erase rejected, to_delete, feature_vec_used)."""
    from ov_plane_amd.build import build_host

    build_host()
    from ov_plane_amd import hostlib

    sc = make_scene(**kw)
    ref = _oracle_full_update(oracle, sc)
    out = hostlib.run_msckf_update(sc)
    assert (out["used"] == ref["used"]).all()
    assert (out["kept"] == ref["kept"]).all()
    assert out["deleted"].all()  # every processed feature is flagged (UpdaterMSCKF.cpp:641,756,791-793)
    assert np.abs(out["clone_p"] - ref["clone_p"]).max() < TOL_DX and np.abs(out["clone_q"] - ref["clone_q"]).max() < TOL_DX
    assert np.abs(out["calib_p"] - ref["calib_p"]).max() < TOL_DX and np.abs(out["intr"] - ref["intr"]).max() < TOL_DX
    if sc.plane_in_state.any():
        assert np.abs(out["cp_state"] - ref["cp"][sc.plane_in_state]).max() < TOL_DX
    assert relP(out["P"], ref["P"]) < TOL_P


def test_rccl_allreduce_path_single_rank(hiplib, oracle):
    """The sharded-update plumbing (zero-copy __cuda_array_interface__ view of the library's [A|b] buffer + RCCL
    all_reduce on the context's stream) with a 1-rank nccl group: results must equal the plain update."""
    import torch
    import torch.distributed as dist

    from ov_plane_amd.dist import sharded_update, shard_bounds

    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29611", rank=0, world_size=1)
    try:
        sc = make_scene(C=9, F=64, seed=81, chi2_mult=1.0)
        ref = oracle.msckf_point_update(sc)
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            ctx = hiplib.Context(sc.N, sc.C, sc.F, stream=stream.cuda_stream)
            ctx.cov_upload(sc.P)
            ctx.state_upload(sc)
            lo, hi = shard_bounds(sc.F, 0, 1)
            ctx.batch_upload_scene(sc, np.arange(lo, hi))
            o = hiplib.opts_from_scene(sc)
            # force the collective even though world_size == 1
            ctx.build_gate_gram_async(o)
            from ov_plane_amd.dist import DeviceBufferView

            ptr, rows, ld = ctx.gram_buffer()
            t = torch.as_tensor(DeviceBufferView(ptr, rows * ld), device="cuda")
            before = t.clone()
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            assert torch.equal(before, t)  # 1 rank: identity, and proves the view aliases the library buffer
            ctx.ekf_update_from_gram_async()
            out = ctx.fetch_results()
            P = ctx.cov_download()
        assert (out["accepted"] == ref["accepted"]).all()
        assert np.abs(out["dx"] - ref["dx"]).max() < TOL_DX and relP(P, ref["P"]) < TOL_P
        # and the packaged helper
        with torch.cuda.stream(stream):
            ctx.cov_upload(sc.P)
            out2 = sharded_update(ctx, o)
        assert np.abs(out2["dx"] - out["dx"]).max() == 0.0
        ctx.close()
        # library-owned streams (what bench.py uses): the collective goes on ovp_ctx_stream through an ExternalStream
        ctx2 = hiplib.Context(sc.N, sc.C, sc.F)
        ctx2.cov_upload(sc.P)
        ctx2.state_upload(sc)
        ctx2.batch_upload_scene(sc)
        ctx2.build_gate_gram_async(o)
        ptr, rows, ld = ctx2.gram_buffer()
        with torch.cuda.stream(torch.cuda.ExternalStream(ctx2.stream_handle())):
            t2 = torch.as_tensor(DeviceBufferView(ptr, rows * ld), device="cuda")
            dist.all_reduce(t2, op=dist.ReduceOp.SUM)
        ctx2.ekf_update_from_gram_async()
        out3 = ctx2.fetch_results()
        assert np.abs(out3["dx"] - out["dx"]).max() == 0.0 and (out3["accepted"] == out["accepted"]).all()
        ctx2.close()
    finally:
        dist.destroy_process_group()


def test_plane_loop_replicated_then_points_sharded(hiplib, oracle):
    """dist.sharded_plane_then_point_update (BASELINE config 4's shape at a size the oracle finishes): plane loop on the whole
    batch, point update on this rank's shard of the free points; with one rank it must equal the plain sequence and the oracle's
    plane loop followed by its point update."""
    import torch

    from ov_plane_amd.dist import sharded_plane_then_point_update

    sc = make_scene(C=12, F=300, seed=19, n_planes=6, feats_per_plane=25, planes_in_state_frac=0.5, chi2_mult=99999.0)
    ref = oracle.msckf_plane_update(sc)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        ctx = hiplib.Context(sc.N, sc.C, sc.F, stream=stream.cuda_stream)
        ctx.cov_upload(sc.P)
        ctx.state_upload(sc)
        o = hiplib.opts_from_scene(sc)
        o2 = hiplib.opts_from_scene(sc)
        o2.chi2_multiplier = 1.0
        pl, pt, mine = sharded_plane_then_point_update(ctx, o, lambda idx: ctx.batch_upload_scene(sc, idx), sc.F,
                                                       (sc.plane_id, sc.cp, sc.cp_fej, sc.plane_state_id), point_opts=o2)
        P1 = ctx.cov_download()
    assert (pl["ok"] == ref["plane_ok"]).all() and (pl["used"] == ref["used"]).all()
    assert (mine == np.nonzero(~ref["used"])[0]).all()
    # the same through the plain calls
    ctx2 = hiplib.Context(sc.N, sc.C, sc.F)
    ctx2.cov_upload(sc.P)
    ctx2.state_upload(sc)
    ctx2.batch_upload_scene(sc)
    pl2 = ctx2.plane_update(o, sc.plane_id, sc.cp, sc.cp_fej, sc.plane_state_id)
    ctx2.batch_upload_scene(sc, np.nonzero(~pl2["used"])[0])
    pt2 = ctx2.msckf_update(o2)
    assert np.abs(pl["dx"] - pl2["dx"]).max() == 0.0 and np.abs(pt["dx"] - pt2["dx"]).max() == 0.0
    # (the sharded call keeps the frame resident: its per-feature results are indexed like the frame)
    assert np.abs(P1 - ctx2.cov_download()).max() == 0.0 and (pt["accepted"][mine] == pt2["accepted"]).all()
    assert not pt["accepted"][pl["used"]].any()
    ctx.close()
    ctx2.close()


@pytest.mark.parametrize("split", ["0", "1"])
@pytest.mark.parametrize("r_iso,chi2_mult,expect", [(1.0, 1e9, 1), (0.25, 1e9, 1), (1.0, 1e-9, 0)])
def test_host_cpp_initialize_matches_oracle(hiplib, oracle, r_iso, chi2_mult, expect, split, monkeypatch):
    """StateHelper::initialize / initialize_invertible (state/StateHelper.cpp:398-586): Givens split, chi2 against the prior,
    covariance augmentation on the device, EKF update with the remaining rows - as one device sequence (ovp_cov_initialize,
    split = 0) and as the three separate calls it replaces (split = 1)."""
    monkeypatch.setenv("OVP_HOST_INIT_SPLIT", split)
    from ov_plane_amd.build import build_host
    from ov_plane_amd.synth import quat_boxplus

    build_host()
    from ov_plane_amd import hostlib

    rng = np.random.default_rng(7)
    sc = make_scene(C=6, F=4, seed=91)
    # the synth layout has calibration at ids 16/22; the host State here does not estimate planes or SLAM features
    order = [(int(sc.ids["clones"][1]), 6), (int(sc.ids["calib"]), 6), (int(sc.ids["clones"][4]), 6)]
    rows, cols, k = 30, 18, 3
    H_R = rng.standard_normal((rows, cols)) * 20.0
    H_L = rng.standard_normal((rows, k)) * 5.0
    res = rng.standard_normal(rows) * np.sqrt(r_iso)
    v0 = np.array([1.0, -2.0, 3.0])
    ref = oracle.initialize(sc.P, order, H_R, H_L, res, r_iso, chi2_mult)
    out = hostlib.run_initialize(sc, order, H_R, H_L, res, r_iso, chi2_mult, v0)
    assert out["ok"] == ref["ok"] == expect
    if not expect:
        return
    assert relP(out["P"], ref["P"]) < TOL_P
    assert np.abs(out["new_value"] - (v0 + ref["new_delta"])).max() < TOL_DX
    dx = ref["dx"]
    for i in range(sc.C):
        cid = sc.ids["clones"][i]
        assert np.abs(out["clone_p"][i] - (sc.clone_p[i] + dx[cid + 3:cid + 6])).max() < TOL_DX
        assert np.abs(out["clone_q"][i] - quat_boxplus(sc.clone_q[i], dx[cid:cid + 3])).max() < TOL_DX


@pytest.mark.parametrize("kw,chi2,on_marginal", [
    # on the marginal of the clone / calibration columns, the rest of the state by the push-through identity (the default) ...
    (dict(C=11, F=140, seed=15, n_planes=3, feats_per_plane=30, planes_in_state_frac=0.0, chi2_mult=1.0), 1e9, True),
    (dict(C=8, F=80, seed=16, n_planes=2, feats_per_plane=25, planes_in_state_frac=0.0, chi2_mult=1.0, ragged=True), 1e9, True),
    # ... and with every kernel on the whole state (OVP_PLANE_INIT_SUB=0)
    (dict(C=11, F=140, seed=15, n_planes=3, feats_per_plane=30, planes_in_state_frac=0.0, chi2_mult=1.0), 1e9, False),
    (dict(C=8, F=80, seed=16, n_planes=2, feats_per_plane=25, planes_in_state_frac=0.0, chi2_mult=1.0, ragged=True), 1e9, False),
    # a state above the register-resident factorizations' 288 columns (N = 300, 306 after the two planes)
    (dict(C=30, F=150, seed=17, n_planes=2, feats_per_plane=35, planes_in_state_frac=0.0, chi2_mult=1.0, n_slam=30), 1e9, True),
])
def test_plane_initialisation_matches_oracle(hiplib, oracle, monkeypatch, kw, chi2, on_marginal):
    """UpdaterPlane::init_vio_plane core (update/UpdaterPlane.cpp:296-481) -> StateHelper::initialize: every accepted plane
    appends 3 columns; state correction, plane value and the augmented covariance against the restatement."""
    if not on_marginal:
        monkeypatch.setenv("OVP_PLANE_INIT_SUB", "0")
    sc = make_scene(**kw)
    if kw.get("n_slam"):
        assert sc.N > 288
    ref = oracle.plane_init(sc, const_init_multi=5.0, const_init_chi2=chi2)
    ctx = hiplib.Context(sc.N + 3 * sc.cp.shape[0], sc.C, sc.F)
    ctx.cov_upload(sc.P)
    ctx.state_upload(sc)
    ctx.batch_upload_scene(sc)
    out = ctx.plane_init(hiplib.opts_from_scene(sc), sc.plane_id, sc.cp, 5.0, chi2)
    assert (out["ok"] == ref["plane_ok"]).all() and out["ok"].all()
    assert (out["new_ids"] == ref["new_id"]).all()
    assert (out["dof"] == ref["plane_dof"]).all()
    assert (out["used"] == ref["used"]).all()
    from ov_plane_amd.synth import quat_boxplus

    cq, cpos, intr = sc.clone_q.copy(), sc.clone_p.copy(), sc.intr.copy()
    cp_end = out["cp"].copy()   # value at initialisation; later planes correct it like any other state variable
    for pl in range(sc.cp.shape[0]):
        dx = out["dx"][pl]
        for g in range(pl):
            cp_end[g] += dx[out["new_ids"][g]:out["new_ids"][g] + 3]
        for i in range(sc.C):
            cid = sc.ids["clones"][i]
            cq[i] = quat_boxplus(cq[i], dx[cid:cid + 3])
            cpos[i] = cpos[i] + dx[cid + 3:cid + 6]
        intr = intr + dx[22:30]
    assert np.abs(cpos - ref["clone_p"]).max() < TOL_DX and np.abs(cq - ref["clone_q"]).max() < TOL_DX
    assert np.abs(intr - ref["intr"]).max() < TOL_DX
    assert np.abs(cp_end - ref["cp"]).max() < TOL_DX
    P = ctx.cov_download()
    assert P.shape == ref["P"].shape
    assert relP(P, ref["P"]) < TOL_P
    ctx.close()


def _apply_dx_to_scene(sc, dx):
    from ov_plane_amd.synth import quat_boxplus

    cq, cp = sc.clone_q.copy(), sc.clone_p.copy()
    for i in range(sc.C):
        cid = sc.ids["clones"][i]
        cq[i] = quat_boxplus(cq[i], dx[cid:cid + 3])
        cp[i] = cp[i] + dx[cid + 3:cid + 6]
    return cq, cp, sc.intr + dx[22:30]


@pytest.mark.parametrize("kw", [
    dict(C=11, n_slam=12, seed=3, outliers=2),
    dict(C=11, n_slam=14, seed=4, n_planes=3, outliers=2, wrong_plane=3),   # plane rows + no-plane fallback
    dict(C=6, n_slam=5, seed=5, do_fej=False),
    dict(C=8, n_slam=8, seed=6, fisheye=True),                               # dense host Jacobian with the equidistant lens
])
@pytest.mark.parametrize("dense", [False, True])
def test_host_cpp_mirror_updater_slam_update(hiplib, oracle, kw, dense):
    """ov_plane::UpdaterSLAM::update (update/UpdaterSLAM.cpp:376-682): per-landmark chi2 over the marginal covariance from the
    device, optional point-on-plane rows with the no-plane fallback, one StateHelper::EKFUpdate on the device.
    dense: the form update() falls back to when the device entry refuses a batch (a track longer than OVP_MAX_MEAS, the gate
    kernel's LDS bound - the reference has no size limit): blocks and gates on the host, marginals from the resident covariance."""
    from ov_plane_amd.build import build_host

    build_host()
    from ov_plane_amd import hostlib
    from ov_plane_amd.synth import make_slam_scene

    sc = make_slam_scene(**kw)
    use_planes = kw.get("n_planes", 0) > 0
    ref = oracle.slam_update(sc, sc.lm_id, use_planes=use_planes)
    hostlib.set_slam_force_dense(dense)
    try:
        out = hostlib.run_updater(sc, "slam_update")
    finally:
        hostlib.set_slam_force_dense(False)
    assert (out["should_marg"] == ~ref["accepted"]).all()
    assert (out["kept"] == ref["accepted"]).all() and out["deleted"].all()
    if use_planes:
        assert ref["fellback"].any()
        # _features_SLAM_to_PLANE: 0 once the plane was dropped for a landmark, the plane id when it was used
        exp = np.where(ref["fellback"], 0, np.where(ref["accepted"], sc.plane_id, -1))
        assert (out["slam_to_plane"] == exp).all()
    cq, cp, intr = _apply_dx_to_scene(sc, ref["dx"])
    assert np.abs(out["clone_p"] - cp).max() < TOL_DX and np.abs(out["clone_q"] - cq).max() < TOL_DX
    assert np.abs(out["intr"] - intr).max() < TOL_DX
    lm = sc.slam_p + ref["dx"][sc.ids["slam"][0]:sc.ids["slam"][0] + 3 * sc.F].reshape(-1, 3)
    assert np.abs(out["slam_p"] - lm).max() < TOL_DX
    assert relP(out["P"], ref["P"]) < TOL_P


@pytest.mark.parametrize("kw", [
    dict(C=11, F=8, seed=5, ragged=True),
    dict(C=8, F=6, seed=6, ragged=True, chi2_mult=0.6),   # two candidates fail the gate
])
@pytest.mark.parametrize("split", ["0", "1"])
def test_host_cpp_mirror_updater_slam_delayed_init(hiplib, oracle, kw, split, monkeypatch):
    """ov_plane::UpdaterSLAM::delayed_init downstream of triangulation (update/UpdaterSLAM.cpp:204-364): one
    StateHelper::initialize per feature, the state grows by 3 for every accepted landmark."""
    monkeypatch.setenv("OVP_HOST_INIT_SPLIT", split)
    from ov_plane_amd.build import build_host

    build_host()
    from ov_plane_amd import hostlib

    sc = make_scene(**kw)
    ref = oracle.slam_delayed_init(sc)
    out = hostlib.run_updater(sc, "slam_delayed_init")
    assert out["n"] == ref["n"]
    assert ((out["new_id"][:sc.F] >= 0) == ref["ok"]).all()
    assert (out["new_id"][:sc.F] == ref["new_id"]).all()
    ok = ref["ok"]
    assert ok.any()
    assert np.abs(out["new_p"][:sc.F][ok] - ref["p"][ok]).max() < TOL_DX
    assert np.abs(out["clone_p"] - ref["clone_p"]).max() < TOL_DX and np.abs(out["clone_q"] - ref["clone_q"]).max() < TOL_DX
    assert np.abs(out["intr"] - ref["intr"]).max() < TOL_DX
    assert relP(out["P"], ref["P"]) < TOL_P
    assert (out["kept"] == ok).all() and out["deleted"].all()


def test_host_cpp_mirror_updater_plane_init(hiplib, oracle):
    """ov_plane::UpdaterPlane::init_vio_plane (update/UpdaterPlane.cpp:296-481) over ovp_plane_init."""
    from ov_plane_amd.build import build_host

    build_host()
    from ov_plane_amd import hostlib

    sc = make_scene(C=11, F=90, seed=33, n_planes=2, feats_per_plane=30, planes_in_state_frac=0.0, chi2_mult=1.0)
    ref = oracle.plane_init(sc, const_init_multi=5.0, const_init_chi2=1.0)
    out = hostlib.run_updater(sc, "plane_init", 5.0, 1.0)
    assert ref["plane_ok"].all()
    assert out["n"] == ref["n"]
    assert (out["new_id"][:2] == ref["new_id"]).all()
    assert np.abs(out["new_p"][:2] - ref["cp"]).max() < TOL_DX
    assert (out["kept"] == ~ref["used"]).all()
    assert (out["deleted"] == ref["used"]).all()
    assert np.abs(out["clone_p"] - ref["clone_p"]).max() < TOL_DX and np.abs(out["clone_q"] - ref["clone_q"]).max() < TOL_DX
    assert np.abs(out["intr"] - ref["intr"]).max() < TOL_DX
    assert relP(out["P"], ref["P"]) < TOL_P


@pytest.mark.parametrize("mode", [dict(use_rk4=1, do_fej=1, imu_avg=0), dict(use_rk4=0, do_fej=0, imu_avg=1),
                                  dict(use_rk4=1, do_fej=1, imu_avg=0, low_rate=True)])
def test_host_cpp_mirror_propagator(hiplib, oracle, mode):
    """ov_plane::Propagator::propagate_and_clone (state/Propagator.cpp:37-126): host Phi/Qd accumulation vs the oracle, then
    StateHelper::EKFPropagation + augment_clone (with the time-offset Jacobian, StateHelper.cpp:613-624) on the device P."""
    from ov_plane_amd.build import build_host

    build_host()
    from ov_plane_amd import hostlib
    from ov_plane_amd.synth import PROP_OPTS, make_imu_scenario
    from oracle import np_ref

    mode = dict(mode)
    low = mode.pop("low_rate", False)
    sc = make_scene(C=6, F=4, seed=91)
    t_off = 0.004
    x, imu, t0, t1 = make_imu_scenario(7, t_state=100.0, dt_cam=0.1, t_off=t_off, low_rate=low)
    po = dict(PROP_OPTS, **mode)
    ref = oracle.propagate_summed(x, po, imu, t0, t1)
    out = hostlib.run_propagate(sc, x, imu, 100.0, 100.1, t_off, po)
    assert np.abs(out["Phi"] - ref["Phi"]).max() < 1e-12
    assert np.abs(out["Q"] - ref["Q"]).max() < 1e-12 * np.abs(ref["Q"]).max()
    assert np.abs(out["last_w"] - ref["last_w"]).max() < 1e-14
    xr = ref["x"]
    x16 = np.concatenate([xr["q"], xr["p"], xr["v"], xr["bg"], xr["ba"]])
    assert np.abs(out["x16"] - x16).max() < 1e-12 and np.abs(out["x16_fej"] - x16).max() < 1e-12
    assert np.abs(out["new_clone"] - x16[:7]).max() < 1e-12
    # covariance: EKFPropagation over the IMU block, clone of the pose, time-offset augmentation
    N = sc.N
    Pp = np_ref.ekf_propagation(sc.P, 0, 15, [(0, 15)], ref["Phi"], ref["Q"])
    Pc = np.zeros((N + 6, N + 6))
    Pc[:N, :N] = Pp
    Pc[N:, :N] = Pp[:6, :]
    Pc[:N, N:] = Pp[:, :6]
    Pc[N:, N:] = Pp[:6, :6]
    dnc = np.concatenate([ref["last_w"], xr["v"]])
    col = Pc[:, 15].copy()
    Pc[:, N:] += np.outer(col, dnc)
    row = Pc[15, :].copy()
    Pc[N:, :] += np.outer(dnc, row)
    assert relP(out["P"], Pc) < 1e-10


def test_host_cpp_mirror_state_maintenance(hiplib, oracle):
    """StateHelper::marginalize_slam (state/StateHelper.cpp:638-652) and merge_planes_and_marginalize (:654-776): landmark
    removal, a plane-merge EKF update (cp_new - cp_old = 0) with its chi2 / angle gate, relabelling of an out-of-state
    target id and removal of unobserved planes, all on the device covariance."""
    from ov_plane_amd.build import build_host

    build_host()
    from ov_plane_amd import hostlib
    from ov_plane_amd.synth import make_slam_scene
    from oracle import np_ref

    sc = make_slam_scene(C=6, n_slam=5, seed=8, n_planes=4)
    # plane 2 is a re-detection of plane 1: same plane up to a small error that is consistent with the covariance
    i1, i2 = int(sc.plane_state_id[0]), int(sc.plane_state_id[1])
    P = sc.P.copy()
    P[i2:i2 + 3, :] = P[i1:i1 + 3, :]
    P[:, i2:i2 + 3] = P[:, i1:i1 + 3]
    P[i2:i2 + 3, i2:i2 + 3] = P[i1:i1 + 3, i1:i1 + 3] + 1e-6 * np.eye(3)
    sc["P"] = P
    sc.cp[1] = sc.cp[0] + np.array([4e-4, -3e-4, 2e-4])
    should_marg = np.array([0, 1, 0, 1, 0], dtype=np.uint8)
    # plane 2 -> 1 (both in state: merge update), plane 3 -> 9 (9 not in state: relabel), plane 4 unobserved (dropped)
    out = hostlib.run_state_maintenance(sc, should_marg, [(2, 1), (3, 9)], [1, 9])
    # ---- expected ----
    keep = np.ones(sc.N, dtype=bool)
    for k in np.where(should_marg)[0]:
        keep[sc.ids["slam"][k]:sc.ids["slam"][k] + 3] = False
    newid = np.cumsum(keep) - 1
    Pe = P[np.ix_(keep, keep)]
    j1, j2 = int(newid[i1]), int(newid[i2])
    wc = 1.0 / 0.001
    H = np.hstack([wc * np.eye(3), -wc * np.eye(3)])
    res = wc * (0.0 - (sc.cp[0] - sc.cp[1]))
    order = [(j1, 3), (j2, 3)]
    Pm = np_ref.get_marginal_covariance(Pe, order)
    S = H @ Pm @ H.T + np.eye(3)
    chi2 = float(res @ np.linalg.solve(S, res))
    assert chi2 < np_ref.chi2_095(3)        # the merge passes its gate in this scenario
    Pe, dx = np_ref.ekf_update(Pe, order, H, res)
    cp1 = sc.cp[0] + dx[j1:j1 + 3]
    cp3 = sc.cp[2] + dx[int(newid[sc.plane_state_id[2]]):int(newid[sc.plane_state_id[2]]) + 3]
    keep2 = np.ones(Pe.shape[0], dtype=bool)
    keep2[j2:j2 + 3] = False
    j4 = int(newid[sc.plane_state_id[3]])
    keep2[j4:j4 + 3] = False
    Pe = Pe[np.ix_(keep2, keep2)]
    assert out["n"] == Pe.shape[0] == sc.N - 6 - 6
    assert relP(out["P"], Pe) < 1e-9
    newid2 = np.cumsum(keep2) - 1
    assert out["plane_id"][0] == newid2[j1] and out["plane_id"][1] == -1 and out["plane_id"][2] == -1
    assert out["plane_id"][3] == -1 and out["plane_id"][8] == newid2[int(newid[sc.plane_state_id[2]])]
    assert np.abs(out["plane_cp"][0] - cp1).max() < TOL_DX and np.abs(out["plane_cp"][8] - cp3).max() < TOL_DX
    assert (out["slam_id"] >= 0).tolist() == [True, False, True, False, True]
    assert out["slam_to_plane"].tolist() == [1, 0, 1, 0, 1]
    exp_ids = [int(newid2[newid[sc.ids["slam"][k]]]) for k in (0, 2, 4)]
    assert out["slam_id"][[0, 2, 4]].tolist() == exp_ids


@pytest.mark.parametrize("kw,refine,one_d", [
    (dict(C=11, F=200, seed=3, ragged=True, min_meas=2), 1, 0),
    (dict(C=30, F=300, seed=4), 1, 0),
    (dict(C=8, F=120, seed=5, ragged=True, min_meas=2), 0, 0),
    (dict(C=11, F=150, seed=6, ragged=True, min_meas=2), 0, 1),     # single_triangulation_1d, no refinement
    (dict(C=30, F=100, seed=7), 1, 1),                               # ... followed by single_gaussnewton
])
def test_triangulation_matches_oracle(hiplib, oracle, kw, refine, one_d):
    """ext FeatureInitializer::single_triangulation + single_gaussnewton on the device against the
    restatement: same features kept / dropped, positions identical up to rounding (sums in the reference's order, no FMA
    contraction, single-precision residuals), and the triangulated batch feeds the update like the scene's own points."""
    sc = make_scene(**kw)
    ref = oracle.triangulate(sc, oracle.triang_defaults(refine_features=refine, triangulate_1d=one_d))
    ctx = hiplib.Context(sc.N, sc.C, sc.F)
    ctx.cov_upload(sc.P)
    ctx.state_upload(sc)
    ctx.batch_upload_scene(sc)
    out = ctx.triangulate(sc.uv_norm, hiplib.triang_defaults(refine_features=refine, triangulate_1d=one_d))
    assert (out["ok"] == ref["ok"]).all()
    # (the 1-d version has no condition-number test: short tracks are not rejected there)
    assert ref["ok"].sum() > 0.8 * sc.F and ((~ref["ok"]).any() or not kw.get("ragged", False) or one_d)
    ok = ref["ok"]
    assert np.abs(out["p_FinG"][ok] - ref["p_FinG"][ok]).max() < 1e-11
    # the linearisation points now on the device are the triangulated ones: update with them == oracle update with them
    from ov_plane_amd.synth import Scene

    sc2 = Scene(sc)
    sc2["p_FinG"] = np.where(ok[:, None], ref["p_FinG"], sc.p_FinG)
    keep = np.where(ok)[0]
    ref_u = oracle.msckf_point_update(sc2, feats=keep)
    # device: same batch, rejected features masked out by a 1-observation count (dropped like UpdaterMSCKF.cpp:94-96)
    ctx.batch_upload_scene(sc2, keep)
    o = hiplib.opts_from_scene(sc)
    upd = ctx.msckf_update(o)
    assert (upd["accepted"] == ref_u["accepted"]).all()
    assert np.abs(upd["dx"] - ref_u["dx"]).max() < TOL_DX
    ctx.close()


def test_host_cpp_mirror_updater_msckf_triangulates_first(hiplib, oracle):
    """UpdaterMSCKF::update handed features WITHOUT positions (uvs_norm only): triangulation + refinement on the device
    (update/UpdaterMSCKF.cpp:120-166), failures erased with to_delete, then the usual update on the survivors."""
    from ov_plane_amd.build import build_host

    build_host(force=False)
    from ov_plane_amd import hostlib
    from ov_plane_amd.synth import Scene, quat_boxplus

    sc = make_scene(C=11, F=150, seed=3, ragged=True, min_meas=2, chi2_mult=1.0)
    tri = oracle.triangulate(sc)
    assert (~tri["ok"]).any()
    keep = np.where(tri["ok"])[0]
    sc2 = Scene(sc)
    sc2["p_FinG"] = np.where(tri["ok"][:, None], tri["p_FinG"], sc.p_FinG)
    ref = oracle.msckf_point_update(sc2, feats=keep)
    out = hostlib.run_msckf_update(sc, triangulate=True)
    exp_kept = np.zeros(sc.F, dtype=bool)
    exp_kept[keep[ref["accepted"]]] = True
    assert (out["kept"] == exp_kept).all()
    assert out["deleted"].all()
    dx = ref["dx"]
    cq, cp = sc.clone_q.copy(), sc.clone_p.copy()
    for i in range(sc.C):
        cid = sc.ids["clones"][i]
        cq[i] = quat_boxplus(cq[i], dx[cid:cid + 3])
        cp[i] = cp[i] + dx[cid + 3:cid + 6]
    assert np.abs(out["clone_p"] - cp).max() < TOL_DX and np.abs(out["clone_q"] - cq).max() < TOL_DX
    assert relP(out["P"], ref["P"]) < TOL_P


def test_closed_loop_over_frames_matches_oracle(hiplib, oracle):
    """Four camera frames in the reference's call order (core/VioManager.cpp:348 propagate_and_clone, :670 UpdaterMSCKF::update,
    :864-866 marginalize_old_clone) through the C++ host mirror with the covariance resident on the device, against the
    same loop composed from the oracle pieces: ids shift after every marginalization, the new clone carries the time-offset
    Jacobian, IMU / dt / calibration / intrinsics receive every update."""
    from ov_plane_amd.build import build_host

    build_host()
    from ov_plane_amd import hostlib
    from ov_plane_amd.synth import (PROP_OPTS, Scene, make_imu_scenario, project_all, quat_2_rot, quat_boxplus,
                                    state_layout)
    from oracle import np_ref

    Cn, K, dtc, t_off = 6, 4, 0.1, 0.004
    sc0 = make_scene(C=Cn, F=4, seed=71)
    rng = np.random.default_rng(5)
    x, _, _, _ = make_imu_scenario(9, t_state=100.0, dt_cam=dtc, t_off=t_off)
    x = dict(x)
    x["q"], x["p"] = sc0.clone_q[-1].copy(), sc0.clone_p[-1].copy()      # the IMU sits at the newest clone
    x["q_fej"], x["p_fej"] = x["q"].copy(), x["p"].copy()
    x["v"] = np.array([0.3, 0.6, 0.0])
    x["v_fej"] = x["v"].copy()
    # one IMU stream for all frames (smooth body rates, specific force ~ gravity in the initial attitude)
    ts = 100.0 + t_off - 0.0113 + np.arange(0, int((K * dtc + 0.04) * 400)) / 400.0
    imu = np.zeros((len(ts), 7))
    imu[:, 0] = ts
    ph = rng.uniform(0, 2 * np.pi, 6)
    for a in range(3):
        imu[:, 1 + a] = 0.2 * np.sin(2 * np.pi * 0.5 * (ts - ts[0]) + ph[a])
        imu[:, 4 + a] = 0.3 * np.sin(2 * np.pi * 0.8 * (ts - ts[0]) + ph[3 + a])
    imu[:, 4:7] += quat_2_rot(x["q"]) @ np.array([0, 0, 9.81])
    po = dict(PROP_OPTS, use_rk4=1, do_fej=1, imu_avg=0)
    frame_time = 100.0 + dtc * np.arange(1, K + 1)
    init = dict(C=Cn, N=sc0.N, clone_q=sc0.clone_q, clone_p=sc0.clone_p, clone_q_fej=sc0.clone_q_fej,
                clone_p_fej=sc0.clone_p_fej, calib_q=sc0.calib_q, calib_p=sc0.calib_p, intr=sc0.intr, x=x, dt=t_off, P=sc0.P,
                t_state=100.0)

    # ---- oracle loop (also generates the measurements of every frame from the state it has reached) ----
    cq, cp = sc0.clone_q.copy(), sc0.clone_p.copy()
    cqf, cpf = sc0.clone_q_fej.copy(), sc0.clone_p_fej.copy()
    calq, calp, intr, dt_est, P = sc0.calib_q.copy(), sc0.calib_p.copy(), sc0.intr.copy(), t_off, sc0.P.copy()
    xs = {k: np.array(v, dtype=np.float64) for k, v in x.items()}
    last_off, t_state, frames = t_off, 100.0, []
    N = sc0.N
    for k in range(K):
        pr = oracle.propagate_summed(xs, po, imu, t_state + last_off, frame_time[k] + dt_est)
        xs = pr["x"]
        P = np_ref.ekf_propagation(P, 0, 15, [(0, 15)], pr["Phi"], pr["Q"])
        Pc = np.zeros((N + 6, N + 6))
        Pc[:N, :N] = P
        Pc[N:, :N] = P[:6, :]
        Pc[:N, N:] = P[:, :6]
        Pc[N:, N:] = P[:6, :6]
        dnc = np.concatenate([pr["last_w"], xs["v"]])
        col = Pc[:, 15].copy()
        Pc[:, N:] += np.outer(col, dnc)
        row = Pc[15, :].copy()
        Pc[N:, :] += np.outer(dnc, row)
        cq, cp = np.vstack([cq, xs["q"]]), np.vstack([cp, xs["p"]])
        cqf, cpf = np.vstack([cqf, xs["q"]]), np.vstack([cpf, xs["p"]])
        last_off, t_state = dt_est, frame_time[k]
        # measurements of this frame: points in front of the middle camera of the window, seen by >= 3 consecutive clones
        Cw, F = Cn + 1, 40
        Rw = np.array([quat_2_rot(q) for q in cq])
        R_ItoC = quat_2_rot(calq)
        mid = Cw // 2
        Rc_mid = R_ItoC @ Rw[mid]
        pc_mid = cp[mid] - Rc_mid.T @ calp
        pts_c = np.stack([rng.uniform(-1.0, 1.0, F), rng.uniform(-0.6, 0.6, F), rng.uniform(2.5, 5.0, F)], axis=1)
        pts = (Rc_mid.T @ pts_c.T).T + pc_mid
        uv_all, z = project_all(pts, Rw, cp, R_ItoC, calp, intr)
        assert (z > 0.5).all()
        nm = rng.integers(3, Cw + 1, size=F).astype(np.int32)
        st = np.array([rng.integers(0, Cw - m + 1) for m in nm])
        uv = np.zeros((F, Cw, 2), dtype=np.float32)
        slot = -np.ones((F, Cw), dtype=np.int32)
        for f in range(F):
            slot[f, :nm[f]] = np.arange(st[f], st[f] + nm[f])
            uv[f, :nm[f]] = (uv_all[f, st[f]:st[f] + nm[f]] + rng.standard_normal((nm[f], 2))).astype(np.float32)
        pf = pts + 0.02 * rng.standard_normal(pts.shape)
        frames.append(dict(uv=uv, slot=slot, n_meas=nm, p_FinG=pf))
        sck = Scene(C=Cw, F=F, N=N + 6, ids=state_layout(Cw), clone_q=cq, clone_p=cp, clone_q_fej=cqf, clone_p_fej=cpf,
                    calib_q=calq, calib_p=calp, intr=intr, P=Pc, uv=uv, clone_idx=slot, n_meas=nm, p_FinG=pf,
                    opts=dict(sigma_px=1.0, sigma_c=0.05, chi2_mult=1.0, do_fej=True, do_calib_pose=True, do_calib_intr=True))
        up = oracle.msckf_point_update(sck)
        dx = up["dx"]
        new = np_ref.apply_dx(sck, dx)
        cq, cp, calq, calp, intr = new["clone_q"], new["clone_p"], new["calib_q"], new["calib_p"], new["intr"]
        xs = dict(xs)
        xs["q"] = quat_boxplus(xs["q"], dx[0:3])
        xs["p"], xs["v"] = xs["p"] + dx[3:6], xs["v"] + dx[6:9]
        xs["bg"], xs["ba"] = xs["bg"] + dx[9:12], xs["ba"] + dx[12:15]
        dt_est = dt_est + dx[15]
        # marginalize the oldest clone
        keep = np.ones(N + 6, dtype=bool)
        keep[30:36] = False
        P = up["P"][np.ix_(keep, keep)]
        cq, cp, cqf, cpf = cq[1:], cp[1:], cqf[1:], cpf[1:]
        frames[-1]["n_acc"] = int(up["accepted"].sum())

    out = hostlib.run_sequence(init, imu, frame_time, frames, po)
    assert out["kept"].tolist() == [fr["n_acc"] for fr in frames] and min(out["kept"]) > 20
    x16 = np.concatenate([xs["q"], xs["p"], xs["v"], xs["bg"], xs["ba"]])
    assert np.abs(out["x16"] - x16).max() < TOL_DX
    assert np.abs(out["clone_p"] - cp).max() < TOL_DX and np.abs(out["clone_q"] - cq).max() < TOL_DX
    assert np.abs(out["calib_p"] - calp).max() < TOL_DX and np.abs(out["intr"] - intr).max() < TOL_DX
    assert abs(out["dt"] - dt_est) < TOL_DX
    assert relP(out["P"], P) < TOL_P


def _golden_mod():
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLD, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    return mg


def test_widened_rows_match_committed_golden_vectors(hiplib):
    """Device results against the committed restatement outputs (tests/golden/wide_*.npz) without the oracle in the loop:
    triangulation, the plane loop and the plane initialisation."""
    mg = _golden_mod()
    # triangulation
    sc = make_scene(**mg.WIDE["triangulate"])
    g = np.load(os.path.join(GOLD, "wide_triangulate.npz"))
    ctx = hiplib.Context(sc.N, sc.C, sc.F)
    ctx.cov_upload(sc.P)
    ctx.state_upload(sc)
    ctx.batch_upload_scene(sc)
    out = ctx.triangulate(sc.uv_norm)
    assert (out["ok"] == g["ok"]).all()
    assert np.abs(out["p_FinG"][g["ok"]] - g["p_FinG"][g["ok"]]).max() < 1e-11
    ctx.close()
    # plane loop
    sc = make_scene(**mg.WIDE["plane_loop"])
    g = np.load(os.path.join(GOLD, "wide_plane_loop.npz"))
    ctx = hiplib.Context(sc.N, sc.C, sc.F)
    ctx.cov_upload(sc.P)
    ctx.state_upload(sc)
    ctx.batch_upload_scene(sc)
    out = ctx.plane_update(hiplib.opts_from_scene(sc), sc.plane_id, sc.cp, sc.cp_fej, sc.plane_state_id)
    assert (out["ok"] == g["plane_ok"]).all() and (out["used"] == g["used"]).all()
    cq, cpos, calq, calp, intr, cp = _apply_plane_dx(sc, out["dx"], out["ok"])
    assert np.abs(cpos - g["clone_p"]).max() < TOL_DX and np.abs(cq - g["clone_q"]).max() < TOL_DX
    assert np.abs(cp - g["cp"]).max() < TOL_DX
    assert relP(ctx.cov_download(), g["P"]) < TOL_P
    ctx.close()
    # plane initialisation
    sc = make_scene(**mg.WIDE["plane_init"])
    g = np.load(os.path.join(GOLD, "wide_plane_init.npz"))
    ctx = hiplib.Context(sc.N + 3 * sc.cp.shape[0], sc.C, sc.F)
    ctx.cov_upload(sc.P)
    ctx.state_upload(sc)
    ctx.batch_upload_scene(sc)
    out = ctx.plane_init(hiplib.opts_from_scene(sc), sc.plane_id, sc.cp, 5.0, 1.0)
    assert (out["ok"] == g["plane_ok"]).all() and (out["new_ids"] == g["new_id"]).all() and (out["used"] == g["used"]).all()
    assert relP(ctx.cov_download(), g["P"]) < TOL_P
    ctx.close()


# ------------------------------------------------------------------------------------------------------------------
# plane fitting: PlaneFitting::plane_fitting / optimize_plane on the device against the restatement
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("variant", [0, 1])
def test_plane_fitting_matches_oracle(hiplib, oracle, variant):
    from ov_plane_amd.synth import make_planefit_problem

    probs = [make_planefit_problem(seed=s, n_feats=nf, outliers=o) for s, nf, o in
             [(1, 24, 3), (2, 40, 6), (3, 12, 0), (4, 9, 2), (5, 64, 10), (6, 130, 20)]]
    pts = [pb["p_FinG"] for pb in probs]
    rng = np.random.default_rng(0)
    pts.append(probs[0]["p_FinG"][:4])                                  # fewer than min_inlier_num (:97-100)
    pts.append(probs[0]["p_FinG"][:1] + 1e-3 * np.arange(8)[:, None])   # no five points 5 cm apart (:138-141)
    pts.append(rng.uniform(-2, 2, (30, 3)) + [0, 0, 5])                 # scattered: no valid inlier set
    fs = np.r_[0, np.cumsum([len(p) for p in pts])]
    ctx = hiplib.Context(32, 2, 4)
    out = ctx.plane_fitting(fs, np.concatenate(pts), 5, 200.0, variant)
    n_ok = 0
    for k, p in enumerate(pts):
        ref = oracle.plane_fitting(p, 5, 200.0, variant)
        assert bool(out["ok"][k]) == ref["ok"], k
        inl = out["inlier"][fs[k]:fs[k + 1]]
        assert (inl == ref["inlier"]).all(), k
        if ref["ok"]:
            n_ok += 1
            assert np.abs(out["abcd"][k] - ref["abcd"]).max() < 1e-9
    assert n_ok == 5  # (4, 9, 2): seven inliers are not more than 80 % of nine
    ctx.close()


def test_plane_optimize_matches_oracle(hiplib, oracle):
    """One launch over a batch of planes: free and fixed planes, SLAM features, a plane that does not converge within the 12
    iterations, one with too few features, one whose estimate is off.  Same success flags and iteration counts as the
    restated Ceres loop, same kept sets, values equal to rounding."""
    from ov_plane_amd.synth import make_planefit_problem

    specs = [dict(seed=11, n_feats=10, n_obs=6), dict(seed=12, n_feats=16, n_obs=8, n_slam=2),
             dict(seed=13, n_feats=8, n_obs=5, fix_plane=True), dict(seed=14, n_feats=5, n_obs=7, n_slam=1, fix_plane=True),
             dict(seed=15, n_feats=30, n_obs=11, n_slam=3), dict(seed=16, n_feats=100, n_obs=11, ragged=True),
             dict(seed=12, n_feats=16, n_obs=8, px_noise=1.0),       # NO_CONVERGENCE
             dict(seed=21, n_feats=3),                               # too few features
             dict(seed=23, n_feats=1, fix_plane=True, cp_noise=0.0, pt_noise=0.005),
             dict(seed=24, n_feats=40, n_obs=9, px_noise=0.5, outliers=4)]
    probs = [make_planefit_problem(**s) for s in specs]
    bad = make_planefit_problem(seed=22, n_feats=10, fix_plane=True)
    bad["cp"] = bad["cp"] * 1.2
    probs.append(bad)
    for pb in probs[1:]:  # the batch shares the current camera pose and the sigmas
        for key in ("R_GtoI", "p_IinG", "R_ItoC", "p_IinC", "sigma_px_norm", "sigma_c"):
            pb[key] = probs[0][key]
    ctx = hiplib.Context(32, 2, 4)
    outs = ctx.plane_optimize(probs)
    n_ok = 0
    for k, (pb, out) in enumerate(zip(probs, outs)):
        ref = oracle.optimize_plane(pb)
        assert out["ok"] == ref["ok"], k
        assert out["iterations"] == ref["iterations"], (k, out["iterations"], ref["iterations"])
        assert (out["kept"] == ref["kept"]).all(), k
        assert np.abs(out["cp"] - ref["cp"]).max() < 1e-9, k
        assert np.abs(out["p_FinG"] - ref["p_FinG"]).max() < 1e-9, k
        n_ok += int(ref["ok"])
    assert n_ok >= 7 and not outs[6]["ok"] and not outs[7]["ok"] and not outs[-1]["ok"]
    ctx.close()


def test_host_cpp_mirror_updater_msckf_fits_planes_first(hiplib, oracle):
    """UpdaterMSCKF::update with nothing pre-computed (update/UpdaterMSCKF.cpp:120-400): triangulation, then for every plane
    the refinement against the in-state plane or RANSAC fit + joint refinement, then the plane loop on the surviving on-plane
    features and the point loop on the rest - composed here from the oracle pieces in the reference's order."""
    from ov_plane_amd.build import build_host

    build_host()
    from ov_plane_amd import hostlib
    from ov_plane_amd.synth import Scene, quat_2_rot

    fit = dict(min_feat=5, max_cond=200.0, variant=0)
    # a filter whose window is accurate relative to the pixel sigma (as after a second of tracking): with reprojection
    # residuals at the Cauchy scale optimize_plane does not converge within its 12 iterations and every plane is skipped
    sc = make_scene(C=10, F=150, seed=72, n_planes=4, feats_per_plane=20, chi2_mult=99999.0, px_noise=0.25, err_scale=0.05)
    tri = oracle.triangulate(sc)
    ok = tri["ok"]
    p_tri = np.where(ok[:, None], tri["p_FinG"], sc.p_FinG)
    R_ItoC, p_IinC = quat_2_rot(sc.calib_q), sc.calib_p
    Rc = np.array([R_ItoC @ quat_2_rot(sc.clone_q[i]) for i in range(sc.C)])
    pc = np.array([sc.clone_p[i] - Rc[i].T @ p_IinC for i in range(sc.C)])
    uvn = np.asarray(sc.uv_norm, dtype=np.float32)
    sc2 = Scene(sc)
    sc2["p_FinG"] = p_tri.copy()
    sc2["plane_id"] = sc.plane_id.copy()
    sc2["cp"] = sc.cp.copy()
    n_est = 0
    for k in range(sc.cp.shape[0]):
        feats = np.where((sc.plane_id == k + 1) & ok)[0]

        def problem(sel, cp, fixp):
            n_obs = sc.n_meas[sel].astype(np.int32)
            rows = [(f, j) for f in sel for j in range(sc.n_meas[f])]
            ci = np.array([sc.clone_idx[f, j] for f, j in rows], dtype=int)
            return dict(n_feats=len(sel), p_FinG=sc2["p_FinG"][sel], n_obs=n_obs,
                        obs_start=np.r_[0, np.cumsum(n_obs)[:-1]].astype(np.int32),
                        uv_norm=np.array([uvn[f, j] for f, j in rows], dtype=np.float64).reshape(-1, 2),
                        R_GtoC=Rc[ci].reshape(-1, 9), p_CinG=pc[ci], cp=cp, fix_plane=fixp,
                        sigma_px_norm=sc.opts["sigma_px"] / sc.intr[0], sigma_c=sc.opts["sigma_c"],
                        R_GtoI=quat_2_rot(sc.clone_q[-1]), p_IinG=sc.clone_p[-1], R_ItoC=R_ItoC, p_IinC=p_IinC)

        keep, cp = None, None
        if sc.plane_in_state[k]:
            res = oracle.optimize_plane(problem(feats, sc.cp[k], True))
            if res["ok"]:
                keep, cp = feats[res["kept"]], sc.cp[k]
        elif len(feats) >= 4:
            fitr = oracle.plane_fitting(sc2["p_FinG"][feats], fit["min_feat"], fit["max_cond"], fit["variant"])
            if fitr["ok"]:
                sel = feats[fitr["inlier"]]
                res = oracle.optimize_plane(problem(sel, -fitr["abcd"][:3] * fitr["abcd"][3], False))
                if res["ok"] and res["n_kept"] >= 4:
                    keep, cp = sel[res["kept"]], res["cp"]
        drop = feats if keep is None else np.setdiff1d(feats, keep)
        sc2["plane_id"][drop] = 0  # not part of the plane update: ordinary MSCKF features of the point loop
        if keep is not None:
            n_est += 1
            sc2["p_FinG"][keep] = res["p_FinG"][res["kept"]]
            if not sc.plane_in_state[k]:
                sc2["cp"][k] = cp
                sc2["cp_fej"][k] = cp
    assert n_est >= 2  # the scenario exercises both kinds of planes
    sc2["plane_id"][~ok] = 0
    good = np.where(ok)[0]
    # plane loop + point loop of the oracle on the surviving features
    sub = Scene(sc2)
    for key in ("uv", "clone_idx", "n_meas", "p_FinG", "plane_id"):
        sub[key] = sc2[key][good]
    sub["F"] = len(good)
    ref = _oracle_full_update(oracle, sub)
    out = hostlib.run_msckf_update(sc, triangulate=True, fit_planes=fit)
    exp_used = np.zeros(sc.F, dtype=bool)
    exp_used[good[ref["used"]]] = True
    exp_kept = np.zeros(sc.F, dtype=bool)
    exp_kept[good[ref["kept"]]] = True
    assert (out["used"] == exp_used).all() and exp_used.sum() > 20
    assert (out["kept"] == exp_kept).all()
    assert np.abs(out["clone_p"] - ref["clone_p"]).max() < TOL_DX and np.abs(out["clone_q"] - ref["clone_q"]).max() < TOL_DX
    assert np.abs(out["cp_state"] - ref["cp"][sc.plane_in_state]).max() < TOL_DX
    assert relP(out["P"], ref["P"]) < TOL_P


def test_host_cpp_mirror_plane_init_fits_planes_first(hiplib, oracle):
    """UpdaterPlane::init_vio_plane with nothing pre-computed (update/UpdaterPlane.cpp:76-481): triangulation of the on-plane
    candidates, RANSAC fit + joint refinement per plane, then the initialisation from the surviving features.  Sixteen
    candidates keep the reference's std::sort by track length a stable insertion sort."""
    from ov_plane_amd.build import build_host

    build_host()
    from ov_plane_amd import hostlib
    from ov_plane_amd.synth import Scene, quat_2_rot

    fit = dict(min_feat=5, max_cond=200.0, variant=0)
    sc = make_scene(C=10, F=16, seed=34, n_planes=2, feats_per_plane=8, planes_in_state_frac=0.0, chi2_mult=1.0,
                    px_noise=0.25, err_scale=0.05)
    assert (sc.plane_id > 0).all()
    tri = oracle.triangulate(sc)
    ok = tri["ok"]
    R_ItoC, p_IinC = quat_2_rot(sc.calib_q), sc.calib_p
    Rc = np.array([R_ItoC @ quat_2_rot(sc.clone_q[i]) for i in range(sc.C)])
    pc = np.array([sc.clone_p[i] - Rc[i].T @ p_IinC for i in range(sc.C)])
    uvn = np.asarray(sc.uv_norm, dtype=np.float32)
    sc2 = Scene(sc)
    sc2["p_FinG"] = np.where(ok[:, None], tri["p_FinG"], sc.p_FinG)
    sc2["plane_id"] = sc.plane_id.copy()
    sc2["cp"] = sc.cp.copy()
    order = np.argsort(sc.n_meas, kind="stable")  # :167-176
    n_est = 0
    for k in range(2):
        feats = np.array([f for f in order if sc.plane_id[f] == k + 1 and ok[f]], dtype=int)
        keep = None
        fitr = oracle.plane_fitting(sc2["p_FinG"][feats], fit["min_feat"], fit["max_cond"], fit["variant"])
        if fitr["ok"]:
            sel = feats[fitr["inlier"]]
            n_obs = sc.n_meas[sel].astype(np.int32)
            rows = [(f, j) for f in sel for j in range(sc.n_meas[f])]
            ci = np.array([sc.clone_idx[f, j] for f, j in rows], dtype=int)
            pb = dict(n_feats=len(sel), p_FinG=sc2["p_FinG"][sel], n_obs=n_obs,
                      obs_start=np.r_[0, np.cumsum(n_obs)[:-1]].astype(np.int32),
                      uv_norm=np.array([uvn[f, j] for f, j in rows], dtype=np.float64).reshape(-1, 2),
                      R_GtoC=Rc[ci].reshape(-1, 9), p_CinG=pc[ci], cp=-fitr["abcd"][:3] * fitr["abcd"][3], fix_plane=False,
                      sigma_px_norm=sc.opts["sigma_px"] / sc.intr[0], sigma_c=sc.opts["sigma_c"],
                      R_GtoI=quat_2_rot(sc.clone_q[-1]), p_IinG=sc.clone_p[-1], R_ItoC=R_ItoC, p_IinC=p_IinC)
            res = oracle.optimize_plane(pb)
            if res["ok"]:
                keep = sel[res["kept"]]
                sc2["p_FinG"][keep] = res["p_FinG"][res["kept"]]
                sc2["cp"][k] = res["cp"]
                sc2["cp_fej"][k] = res["cp"]
                n_est += 1
        drop = np.where(sc.plane_id == k + 1)[0] if keep is None else np.setdiff1d(np.where(sc.plane_id == k + 1)[0], keep)
        sc2["plane_id"][drop] = 0
    assert n_est == 2
    ref = oracle.plane_init(sc2, const_init_multi=5.0, const_init_chi2=1.0)
    out = hostlib.run_updater(sc, "plane_init", 5.0, 1.0, fit_planes=fit)
    assert ref["plane_ok"].all()
    assert out["n"] == ref["n"]
    assert (out["new_id"][:2] == ref["new_id"]).all()
    assert np.abs(out["new_p"][:2] - ref["cp"]).max() < TOL_DX
    assert (out["deleted"] == ref["used"]).all()
    assert np.abs(out["clone_p"] - ref["clone_p"]).max() < TOL_DX and np.abs(out["clone_q"] - ref["clone_q"]).max() < TOL_DX
    assert relP(out["P"], ref["P"]) < TOL_P


# ------------------------------------------------------------------------------------------------------------------
# on-disk formats
# ------------------------------------------------------------------------------------------------------------------
def test_committed_trace_frame_replays_on_the_device(hiplib):
    """tests/golden/trace_c6.ovptrc: a frame in the binary trace format with the oracle's outputs; replaying its inputs
    through the C-ABI gives those outputs (the offline comparison a frame recorded next to the reference would get)."""
    from ov_plane_amd import trace

    f = trace.read_frames(os.path.join(GOLD, "trace_c6.ovptrc"))[0]
    sc = trace.scene_from_frame(f)
    out = run_gpu(hiplib, sc)
    assert (out["accepted"] == f["accepted"].astype(bool)).all()
    assert np.abs(out["chi2"] - f["chi2"]).max() <= 1e-8 * max(1.0, np.abs(f["chi2"]).max())
    assert np.abs(out["dx"] - f["dx"]).max() < TOL_DX
    assert relP(out["P"], f["P_after"]) < TOL_P
    out["ctx"].close()


def test_euroc_sized_trace_replays_on_the_device(hiplib, oracle):
    """tests/golden/trace_euroc_like.ovptrc (see tests/test_formats_cpu.py): replaying the recorded frames through the C-ABI
    reproduces the recorded device outputs and agrees with the oracle on every frame (11 + 1 clones, <= 20 features, gate at
    chi2_multipler = 1: the small-batch regime of config/euroc_mav/estimator_config.yaml)."""
    from ov_plane_amd import trace

    path = os.path.join(GOLD, "trace_euroc_like.ovptrc")
    rows = trace.replay(path)
    assert len(rows) >= 8
    for r in rows:
        assert r["accept_mismatch"] == 0 and r["max_abs_ddx"] < 1e-9 and r["max_rel_dP"] < 1e-8, r
    f = trace.read_frames(path)[-1]
    sc = trace.scene_from_frame(f)
    out = run_gpu(hiplib, sc)
    ref = oracle.msckf_point_update(sc)
    assert (out["accepted"] == ref["accepted"]).all()
    assert np.abs(out["dx"] - ref["dx"]).max() < TOL_DX and relP(out["P"], ref["P"]) < TOL_P
    out["ctx"].close()


@pytest.mark.parametrize("with_gt", [0, 1])
def test_state_files_match_the_reference_layout(hiplib, with_gt):
    """ROSVisualizerHelper::sim_save_total_state_to_file (ros/ROSVisualizerHelper.cpp:152-302): one line of the estimate,
    standard-deviation and groundtruth files each, with the reference's precisions (5 / 6 / 7 / 0 digits)."""
    import ctypes as C

    from ov_plane_amd.build import build_host

    build_host()
    from ov_plane_amd import hostlib

    L = hostlib.lib()
    sc = make_scene(C=5, F=4, seed=9)
    f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)  # noqa: E731
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    cq, cp, calq, calp, intr, P = f64(sc.clone_q), f64(sc.clone_p), f64(sc.calib_q), f64(sc.calib_p), f64(sc.intr), np.asfortranarray(sc.P)
    est, sd, gt = (C.create_string_buffer(4096) for _ in range(3))
    ts, dt = 1403715273.262142, 0.0041234567
    rc = L.ovph_format_state_files(C.c_int(sc.C), p(cq), p(cp), p(calq), p(calp), p(intr), C.c_int(sc.N), p(P), C.c_double(ts),
                                   C.c_double(dt), C.c_int(with_gt), est, sd, gt, C.c_int(4096))
    assert rc == 0
    gt_dt = 0.0123456
    t_out = ts + (gt_dt if with_gt else dt)  # :160 / :169: the groundtruth offset stamps all three files in simulation
    six = lambda v: " ".join("%.6f" % x for x in v)  # noqa: E731
    # the IMU sits at the last clone that was pushed; velocity and biases are the State's initial zeros
    imu = np.r_[sc.clone_q[-1], sc.clone_p[-1], np.zeros(9)]
    exp_est = "%.5f %s %.7f 1 %s %s \n" % (t_out, six(imu), dt, six(sc.intr), six(np.r_[sc.calib_q, sc.calib_p]))
    assert est.value.decode() == exp_est
    s = np.sqrt(np.diag(sc.P))
    ids = sc.ids
    exp_sd = "%.5f %s %.6f 1 %s %s \n" % (t_out, six(s[0:15]), s[15], six(s[ids["intr"]:ids["intr"] + 8]),
                                         six(s[ids["calib"]:ids["calib"] + 6]))
    assert sd.value.decode() == exp_sd
    if with_gt:
        g = 0.5 + 0.125 * np.arange(17)
        exp_gt = "%.5f %s %.7f 1 %s %s \n" % (g[0], six(g[1:]), gt_dt, six(sc.intr + 1.0), six(0.1 * np.arange(1, 8)))
        assert gt.value.decode() == exp_gt
    else:
        assert gt.value.decode() == ""


def test_host_cpp_mirror_updater_msckf_with_slam_landmarks_on_planes(hiplib, oracle):
    """UpdaterMSCKF::update on a state that holds SLAM landmarks lying on planes which are not in the state
    (update/UpdaterMSCKF.cpp:232-252): the landmark takes part in the RANSAC fit and in the refinement as a constant, and in
    the plane update with one constraint row on its own state columns; one landmark per plane keeps the reference's
    unordered_map iteration order out of the picture."""
    from ov_plane_amd.build import build_host

    build_host()
    from ov_plane_amd import hostlib
    from ov_plane_amd.synth import Scene, quat_2_rot, slam_rows_on_planes

    fit = dict(min_feat=5, max_cond=200.0, variant=0)
    sc = make_scene(C=10, F=150, seed=72, n_planes=4, feats_per_plane=20, n_slam=2, chi2_mult=99999.0, px_noise=0.25,
                    err_scale=0.05)
    slam = slam_rows_on_planes(sc, 2)
    assert sorted(slam["plane"].tolist()) == [3, 4] and list(sc.plane_in_state) == [True, True, False, False]
    tri = oracle.triangulate(sc)
    ok = tri["ok"]
    R_ItoC, p_IinC = quat_2_rot(sc.calib_q), sc.calib_p
    Rc = np.array([R_ItoC @ quat_2_rot(sc.clone_q[i]) for i in range(sc.C)])
    pc = np.array([sc.clone_p[i] - Rc[i].T @ p_IinC for i in range(sc.C)])
    uvn = np.asarray(sc.uv_norm, dtype=np.float32)
    sc2 = Scene(sc)
    sc2["p_FinG"] = np.where(ok[:, None], tri["p_FinG"], sc.p_FinG)
    sc2["plane_id"] = sc.plane_id.copy()
    sc2["cp"] = sc.cp.copy()
    slam_kept = []
    for k in range(4):
        feats = np.where((sc.plane_id == k + 1) & ok)[0]
        lms = [q for q in range(2) if slam["plane"][q] == k + 1] if not sc.plane_in_state[k] else []

        def problem(sel, lm_sel, cp, fixp):
            n_obs = np.r_[sc.n_meas[sel], np.zeros(len(lm_sel))].astype(np.int32)
            rows = [(f, j) for f in sel for j in range(sc.n_meas[f])]
            ci = np.array([sc.clone_idx[f, j] for f, j in rows], dtype=int)
            pts = np.vstack([sc2["p_FinG"][sel]] + [slam["p"][q][None] for q in lm_sel])
            return dict(n_feats=len(sel) + len(lm_sel), p_FinG=pts, n_obs=n_obs,
                        obs_start=np.r_[0, np.cumsum(n_obs)[:-1]].astype(np.int32),
                        uv_norm=np.array([uvn[f, j] for f, j in rows], dtype=np.float64).reshape(-1, 2),
                        R_GtoC=Rc[ci].reshape(-1, 9), p_CinG=pc[ci], cp=cp, fix_plane=fixp,
                        sigma_px_norm=sc.opts["sigma_px"] / sc.intr[0], sigma_c=sc.opts["sigma_c"],
                        R_GtoI=quat_2_rot(sc.clone_q[-1]), p_IinG=sc.clone_p[-1], R_ItoC=R_ItoC, p_IinC=p_IinC)

        keep, refined, kept_lm = None, None, []
        if sc.plane_in_state[k]:
            res = oracle.optimize_plane(problem(feats, [], sc.cp[k], True))
            if res["ok"]:
                keep, refined = feats[res["kept"]], res["p_FinG"][res["kept"]]
        else:
            pts = np.vstack([sc2["p_FinG"][feats]] + [slam["p"][q][None] for q in lms])
            fitr = oracle.plane_fitting(pts, fit["min_feat"], fit["max_cond"], fit["variant"])
            if fitr["ok"]:
                sel = feats[fitr["inlier"][:len(feats)]]
                lm_sel = [q for j, q in enumerate(lms) if fitr["inlier"][len(feats) + j]]
                res = oracle.optimize_plane(problem(sel, lm_sel, -fitr["abcd"][:3] * fitr["abcd"][3], False))
                if res["ok"] and res["n_kept"] >= 4:
                    km = res["kept"][:len(sel)]
                    keep, refined = sel[km], res["p_FinG"][:len(sel)][km]
                    kept_lm = [q for j, q in enumerate(lm_sel) if res["kept"][len(sel) + j]]
                    sc2["cp"][k] = res["cp"]
                    sc2["cp_fej"][k] = res["cp"]
        drop = feats if keep is None else np.setdiff1d(feats, keep)
        sc2["plane_id"][drop] = 0
        if keep is not None:
            sc2["p_FinG"][keep] = refined
            slam_kept += kept_lm
    assert len(slam_kept) == 2  # both landmarks survive fit and refinement
    sc2["plane_id"][~ok] = 0
    good = np.where(ok)[0]
    sub = Scene(sc2)
    for key in ("uv", "clone_idx", "n_meas", "p_FinG", "plane_id"):
        sub[key] = sc2[key][good]
    sub["F"] = len(good)
    slam_sel = dict(plane=slam["plane"][slam_kept], id=slam["id"][slam_kept], p=slam["p"][slam_kept], p_fej=slam["p_fej"][slam_kept])
    ref = _oracle_full_update(oracle, sub, slam=slam_sel)
    out = hostlib.run_updater(sc, "msckf_fit", fit_planes=fit, slam=slam)
    exp_kept = np.zeros(sc.F, dtype=bool)
    exp_kept[good[ref["kept"]]] = True
    assert (out["kept"] == exp_kept).all()
    assert np.abs(out["clone_p"] - ref["clone_p"]).max() < TOL_DX and np.abs(out["clone_q"] - ref["clone_q"]).max() < TOL_DX
    assert np.abs(out["cp"] - ref["cp"][sc.plane_in_state]).max() < TOL_DX
    assert np.abs(out["slam_p"][slam_kept] - ref["slam_p"]).max() < TOL_DX
    assert relP(out["P"], ref["P"]) < TOL_P
    assert list(out["slam_to_plane"][:2]) == [int(slam["plane"][0]), int(slam["plane"][1])]  # :636-639


@pytest.mark.parametrize("do_fej", [False, True])
def test_host_cpp_mirror_landmark_representations(hiplib, oracle, do_fej):
    """UpdaterHelper::get_feature_jacobian_representation / get_feature_jacobian_full (C++ host mirror) for every ext
    LandmarkRepresentation against the C restatement (itself pinned by finite differences): the dense path UpdaterSLAM uses."""
    from ov_plane_amd.build import build_host

    build_host()
    from ov_plane_amd import hostlib

    sc = make_scene(C=7, F=4, seed=5, do_fej=do_fej, ragged=True)
    for f in range(sc.F):
        anchor = int(sc.clone_idx[f, min(1, sc.n_meas[f] - 1)])
        for rep in range(6):
            H_f, H_x, r, order = hostlib.feature_jacobian_rep(sc, f, rep, anchor)
            R_f, R_x, rr, rorder = oracle.feature_jacobian_full_rep(sc, f, rep, anchor)
            assert order == rorder and H_f.shape == R_f.shape, (f, rep)
            assert np.abs(r - rr).max() < 1e-9
            assert np.abs(H_f - R_f).max() < 1e-9 * max(1.0, np.abs(R_f).max()), (f, rep)
            assert np.abs(H_x - R_x).max() < 1e-9 * max(1.0, np.abs(R_x).max()), (f, rep)


@pytest.mark.parametrize("rep,do_fej", [(2, True), (3, True), (4, True), (4, False)])
def test_host_cpp_mirror_change_anchors(hiplib, oracle, rep, do_fej):
    """UpdaterSLAM::change_anchors / perform_anchor_change (update/UpdaterSLAM.cpp:684-850) on the device-resident covariance
    against the restatement (which is pinned by the invariance of the landmark's global error covariance)."""
    from ov_plane_amd.build import build_host

    build_host()
    from ov_plane_amd import hostlib
    from ov_plane_amd.synth import quat_2_rot

    sc = make_scene(C=6, F=4, seed=5, n_slam=2, do_fej=do_fej)
    lm_id = int(sc.ids["slam"][0])
    R_ItoC, p_IinC = quat_2_rot(sc.calib_q), sc.calib_p
    p_G = sc.p_FinG[0]
    p_A = R_ItoC @ quat_2_rot(sc.clone_q[0]) @ (p_G - sc.clone_p[0]) + p_IinC
    p_A_fej = p_A + np.array([2e-3, -1e-3, 3e-3]) * do_fej
    ref = oracle.anchor_change(sc, rep, 0, sc.C - 1, lm_id, p_A, p_A_fej)
    out = hostlib.run_change_anchors(sc, rep, p_A, p_A_fej)
    assert out["anchor_ci"] == sc.C - 1

    def params(p):  # representation parameters of an anchor-frame position (ext Landmark::set_from_xyz)
        if rep == 2:
            return p
        if rep == 3:
            rho = 1 / np.linalg.norm(p)
            return np.array([np.arctan2(p[1], p[0]), np.arccos(rho * p[2]), rho])
        return np.array([p[0] / p[2], p[1] / p[2], 1 / p[2]])

    assert np.abs(out["value"] - params(ref["p_FinA"])).max() < 1e-10
    assert np.abs(out["fej"] - params(ref["p_FinA_fej"])).max() < 1e-10
    assert relP(out["P"], ref["P"]) < TOL_P
    assert np.abs(out["P"] - ref["P"]).max() < 1e-9 * np.abs(ref["P"]).max()


@pytest.mark.parametrize("rep", [1, 2, 3, 4])
def test_host_cpp_mirror_updater_slam_update_with_landmark_representations(hiplib, oracle, rep):
    """UpdaterSLAM::update with the landmarks held in an inverse-depth / anchored representation: dense Jacobians with the
    representation and anchor terms on the host (update/UpdaterHelper.cpp:35-193, :411-421), EKF update on the device; the
    correction of a landmark is one of its representation parameters."""
    from ov_plane_amd.build import build_host

    build_host()
    from ov_plane_amd import hostlib
    from ov_plane_amd.synth import make_slam_scene, quat_2_rot

    sc = make_slam_scene(C=8, n_slam=8, seed=6, outliers=1)
    anchor = 2
    ref = oracle.slam_update(sc, sc.lm_id, rep=np.full(sc.F, rep), anchor=np.full(sc.F, anchor))
    assert ref["rc"] == 0 and ref["accepted"].sum() >= 6 and not ref["accepted"].all()
    out = hostlib.run_updater(sc, "slam_update", slam_rep=(rep, anchor))
    assert (out["kept"] == ref["accepted"]).all()
    dx = ref["dx"]
    R_ItoC, p_IinC = quat_2_rot(sc.calib_q), sc.calib_p

    def params(p_G):
        p = p_G if rep == 1 else R_ItoC @ quat_2_rot(sc.clone_q[anchor]) @ (p_G - sc.clone_p[anchor]) + p_IinC
        if rep in (1, 3):
            rho = 1 / np.linalg.norm(p)
            return np.array([np.arctan2(p[1], p[0]), np.arccos(rho * p[2]), rho])
        if rep == 2:
            return p
        return np.array([p[0] / p[2], p[1] / p[2], 1 / p[2]])

    for k in range(sc.F):
        i = int(sc.lm_id[k])
        assert np.abs(out["slam_p"][k] - (params(sc.slam_p[k]) + dx[i:i + 3])).max() < TOL_DX, k
    cq, cp, intr = _apply_dx_to_scene(sc, dx)
    assert np.abs(out["clone_p"] - cp).max() < TOL_DX and np.abs(out["clone_q"] - cq).max() < TOL_DX
    assert np.abs(out["intr"] - intr).max() < TOL_DX
    assert relP(out["P"], ref["P"]) < TOL_P


@pytest.mark.parametrize("rep", [1, 2, 3, 4, 5])
def test_host_cpp_mirror_updater_slam_delayed_init_with_feat_rep_slam(hiplib, oracle, rep):
    """UpdaterSLAM::delayed_init with StateOptions::feat_rep_slam != GLOBAL_3D (update/UpdaterSLAM.cpp:230-296): the landmark
    joins the state in its representation, anchored in the camera of its last measurement; the single inverse depth drops
    its bearing columns by a nullspace projection and adds ONE column to the state."""
    from ov_plane_amd.build import build_host

    build_host()
    from ov_plane_amd import hostlib

    sc = make_scene(C=8, F=6, seed=6, ragged=True, chi2_mult=0.6)
    ref = oracle.slam_delayed_init(sc, rep=rep)
    out = hostlib.run_updater(sc, "slam_delayed_init", feat_rep_slam=rep)
    k = 1 if rep == 5 else 3
    ok = ref["ok"]
    assert ok.any() and not ok.all()
    assert out["n"] == ref["n"] == sc.N + k * ok.sum()
    assert (out["new_id"][:sc.F] == ref["new_id"]).all()
    assert np.abs(out["new_p"][:sc.F][ok][:, :k] - ref["p"][ok][:, :k]).max() < TOL_DX
    assert np.abs(out["clone_p"] - ref["clone_p"]).max() < TOL_DX and np.abs(out["clone_q"] - ref["clone_q"]).max() < TOL_DX
    assert np.abs(out["intr"] - ref["intr"]).max() < TOL_DX
    assert relP(out["P"], ref["P"]) < TOL_P
    assert (out["kept"] == ok).all() and out["deleted"].all()


@pytest.mark.parametrize("case", ["standing", "moving", "moving_low_disparity", "two_frames"])
def test_host_cpp_mirror_updater_zero_velocity(hiplib, oracle, case):
    """ov_plane::UpdaterZeroVelocity::try_update (update/UpdaterZeroVelocity.cpp:68-318): detection on the host from the
    9 x 9 marginal of the device covariance, bias walk (StateHelper::EKFPropagation) and the stacked IMU rows
    (StateHelper::EKFUpdate, non-isotropic diagonal R) on the device; state time moves on without a clone; on the second
    consecutive acceptance the tracks lose the measurements of the previous zero-velocity time (:245-247)."""
    from ov_plane_amd.build import build_host

    build_host()
    from ov_plane_amd import hostlib
    from ov_plane_amd.synth import PROP_OPTS, make_imu_scenario

    sc = make_scene(C=6, F=4, seed=91)
    po = dict(PROP_OPTS)
    t_off, t_state = 0.004, 100.0
    standing = case in ("standing", "two_frames")
    n_cam = 2 if case == "two_frames" else 1
    x, imu, t0, t1 = make_imu_scenario(5, t_state=t_state, t_off=t_off, stationary=standing, n_cam=n_cam)
    stamps = [t_state + 0.1 * (k + 1) for k in range(n_cam)]
    rng = np.random.default_rng(3)
    uv0 = rng.uniform(50, 700, (30, 2)).astype(np.float32)
    shift = 0.2 if case == "moving_low_disparity" else 6.0   # pixels between the two images
    uv1 = (uv0 + shift * np.array([0.6, 0.8])).astype(np.float32)
    out = hostlib.run_zupt(sc, x, imu, t_state, stamps, t_off, po, uv0=uv0, uv1=uv1)
    disparity_passed = case == "moving_low_disparity"
    ref = oracle.zupt_update(x, po, sc.P, imu, t0, t1, disparity_passed=disparity_passed)
    assert out["accepted"][0] == ref["accepted"] == (case != "moving")
    assert abs(out["chi2"][0] - ref["chi2"]) < 1e-8 * max(1.0, ref["chi2"])
    if not ref["accepted"]:
        assert out["timestamp"] == t_state and np.abs(out["P"] - sc.P).max() < 1e-15 and out["meas_at_t1"] == 30
        return
    dx, P, xr, dt_new = ref["dx"], ref["P"], x, t_off
    from ov_plane_amd.synth import quat_boxplus

    def _apply_imu_dx(xa, d):
        return dict(xa, q=quat_boxplus(xa["q"], d[0:3]), p=xa["p"] + d[3:6], v=xa["v"] + d[6:9], bg=xa["bg"] + d[9:12],
                    ba=xa["ba"] + d[12:15])

    xr = _apply_imu_dx(x, dx)
    dt_new = t_off + dx[15]
    if n_cam == 2:
        # second frame: time0 = previous camera time + the PREVIOUS offset, time1 = new camera time + the corrected offset
        ref2 = oracle.zupt_update(xr, po, P, imu, stamps[0] + t_off, stamps[1] + dt_new)
        assert ref2["accepted"] and out["accepted"][1]
        assert abs(out["chi2"][1] - ref2["chi2"]) < 1e-7 * max(1.0, ref2["chi2"])
        xr = _apply_imu_dx(xr, ref2["dx"])
        dt_new += ref2["dx"][15]
        P = ref2["P"]
        assert out["meas_at_t1"] == 0          # cleanup_measurements_exact(last_zupt_state_timestamp)
    else:
        assert out["meas_at_t1"] == 30
    assert out["timestamp"] == stamps[-1]
    x16 = np.concatenate([xr["q"], xr["p"], xr["v"], xr["bg"], xr["ba"]])
    assert np.abs(out["x16"] - x16).max() < TOL_DX and abs(out["calib_dt"] - dt_new) < TOL_DX
    assert relP(out["P"], P) < TOL_P


@pytest.mark.parametrize("planes", [0, 1, 2])
def test_closed_loop_vio_on_simulated_data_is_accurate_and_consistent(hiplib, planes):
    """the restated Simulator (ov_plane_amd/sim.py) drives propagate -> triangulate -> MSCKF update ->
    marginalise through the C++ host mirror with the covariance on the device for 12 s of motion.  No oracle here: the filter is
    scored against the simulator's ground truth - drift stays at the centimetre level and the NEES of position and orientation
    stays around its expectation of 3 (the estimator neither diverges nor becomes overconfident).  planes = 1: the simulator's
    point-to-plane associations arrive as feat2plane and UpdaterMSCKF uses the planar regularities; planes = 2: additionally
    UpdaterPlane::init_vio_plane puts planes into the state (core/VioManager.cpp:583-588)."""
    from ov_plane_amd.build import build_host

    build_host()
    from ov_plane_amd import closed_loop
    from ov_plane_amd.sim import Simulator, synthetic_trajectory

    sim = Simulator(synthetic_trajectory(duration=30.0), num_pts=100, num_pts_plane=100)
    r = closed_loop.run(sim, n_frames=120, C=11, planes=planes)
    assert (r["planes_in_state"] >= 1) == (planes == 2) and (r["planar_per_frame"].sum() > 0) == (planes > 0)
    assert r["feats_per_frame"].sum() > 5 * 120, r["feats_per_frame"]
    assert r["kept_per_frame"].sum() >= 0.7 * r["feats_per_frame"].sum(), (r["kept_per_frame"], r["feats_per_frame"])
    assert r["rmse_pos"] < 0.15 and r["e_pos"].max() < 0.3, (r["rmse_pos"], r["e_pos"].max())
    assert r["rmse_ori_deg"] < 0.3, r["rmse_ori_deg"]
    assert 0.2 < r["nees_pos"].mean() < 9.0 and 0.2 < r["nees_ori"].mean() < 9.0, (r["nees_pos"].mean(), r["nees_ori"].mean())
    if planes == 2:
        return    # the covariance grew by the planes and is not returned
    P = r["final"]["P"]
    # the newest clone is a copy of the IMU pose: P is positive SEMI-definite by construction
    w = np.linalg.eigvalsh(P)
    assert np.abs(P - P.T).max() < 1e-12 * np.abs(P).max() and w.min() > -1e-12 * w.max()


@pytest.mark.parametrize("planes,max_slam,rep", [(0, 0, 0), (2, 0, 0), (0, 25, 0), (2, 25, 0), (0, 25, 2), (0, 25, 4), (0, 25, 5)])
def test_filter_session_with_slam_landmarks_planes_and_anchor_changes(hiplib, tmp_path, planes, max_slam, rep):
    """csrc/host/ov_plane_session.cpp: the VioManager slice (propagate -> marginalise lost landmarks -> plane init -> MSCKF update
    -> SLAM update -> delayed init -> anchor change -> marginalise the oldest clone) frame by frame with the covariance resident
    on the device, driven by the simulator.  Long tracks become SLAM landmarks (GLOBAL_3D, ANCHORED_3D, ANCHORED_MSCKF_INVERSE_DEPTH
    and the single inverse depth - the anchored ones change their anchor every time its clone leaves the window), lost ones are
    marginalised; accuracy and consistency against the simulator's ground truth."""
    from ov_plane_amd.build import build_host

    build_host()
    from ov_plane_amd import closed_loop
    from ov_plane_amd.sim import Simulator, synthetic_trajectory

    sim = Simulator(synthetic_trajectory(duration=30.0), num_pts=100, num_pts_plane=100)
    out_dir = str(tmp_path) if (planes, max_slam, rep) == (2, 25, 0) else None
    r = closed_loop.run_session(sim, n_frames=120, C=11, planes=planes, max_slam=max_slam, feat_rep_slam=rep, out_dir=out_dir)
    c = r["counts"]
    if out_dir:
        # the files a run of the reference's simulation leaves behind: one line per frame, groundtruth and estimate aligned
        est = np.loadtxt(os.path.join(out_dir, "state_estimate.txt"))
        std = np.loadtxt(os.path.join(out_dir, "state_deviation.txt"))
        gt = np.loadtxt(os.path.join(out_dir, "state_groundtruth.txt"))
        assert est.shape[0] == std.shape[0] == gt.shape[0] == 120 and est.shape[1] == gt.shape[1]
        assert np.abs(est[:, 0] - r["times"]).max() < 5e-3 and np.abs(est[:, 1:8] - r["traj"][:, 0:7]).max() < 1e-5
        assert np.abs(np.linalg.norm(est[:, 5:8] - gt[:, 5:8], axis=1) - r["e_pos"]).max() < 1e-5
        assert (std[:, 1:] >= 0).all() and np.abs(std[:, 4:7] - np.sqrt(np.stack([np.diag(P)[3:6] for P in r["posecov"]]))).max() < 1e-5
        with open(os.path.join(out_dir, "timing.txt")) as fh:
            lines = fh.read().strip().split("\n")
        assert lines[0].startswith("#") and "slam update" in lines[0] and len(lines) == 121
        tm = np.array([[float(v) for v in ln.split(",")] for ln in lines[1:]])
        assert (tm[:, 1:] >= 0).all() and np.abs(tm[:, 1:-1].sum(axis=1) - tm[:, -1]).max() < 2e-3
    assert r["rmse_pos"] < 0.15 and r["e_pos"].max() < 0.3, (r["rmse_pos"], r["e_pos"].max())
    assert r["rmse_ori_deg"] < 0.3, r["rmse_ori_deg"]
    assert 0.2 < r["nees_pos"].mean() < 9.0 and 0.2 < r["nees_ori"].mean() < 9.0, (r["nees_pos"].mean(), r["nees_ori"].mean())
    if max_slam:
        assert c[:, 4].max() <= max_slam and c[5:, 4].min() >= 10        # the landmark set stays populated and bounded
        assert c[:, 1].sum() > 15 * 100 and c[:, 2].sum() > 40 and c[:, 3].sum() > 30   # updates, initialisations, marginalisations
    else:
        assert c[:, 1:5].sum() == 0
    if planes == 2 and not max_slam:
        assert c[-1, 5] >= 1
    if planes == 0:
        assert c[:, 5].sum() == 0


def test_filter_session_on_the_reference_dataset_excerpt(hiplib):
    """The reference's default simulation dataset (first 60 s of data/udel_arl_short.txt, committed under tests/golden/) through
    the C++ trajectory loader, the restated simulator and the filter session with SLAM landmarks and planes: 30 s of estimation
    at the dataset's 1.2 m/s."""
    from ov_plane_amd.build import build_host

    build_host()
    from ov_plane_amd import closed_loop, hostlib
    from ov_plane_amd.sim import Simulator

    traj = hostlib.load_trajectory(os.path.join(GOLD, "udel_arl_short_60s.txt"))
    sim = Simulator(traj, num_pts=100, num_pts_plane=100)
    r = closed_loop.run_session(sim, n_frames=300, C=11, planes=2, max_slam=25)
    assert r["rmse_pos"] < 0.3 and r["rmse_ori_deg"] < 0.5, (r["rmse_pos"], r["rmse_ori_deg"])
    assert 0.2 < r["nees_pos"].mean() < 9.0 and 0.2 < r["nees_ori"].mean() < 9.0, (r["nees_pos"].mean(), r["nees_ori"].mean())
    assert r["counts"][:, 1].sum() > 10 * 300


def test_filter_session_zero_velocity_updates_while_standing(hiplib):
    """VioManagerOptions::try_zupt in the session (core/VioManager.cpp:311-331): the platform stands still for the first seconds;
    every such frame is recognised by UpdaterZeroVelocity (chi2 of the raw IMU readings / disparity of the tracks), gets a
    zero-velocity update on the device instead of a clone, and the estimate does not move; once it accelerates the detector
    lets go and the normal pipeline (MSCKF + SLAM) takes over.  (No NEES claim for the start from rest: the first updates
    triangulate over clones without parallax, with or without the zero-velocity updates.)"""
    from ov_plane_amd.build import build_host

    build_host()
    from ov_plane_amd import closed_loop
    from ov_plane_amd.sim import Simulator, synthetic_trajectory

    sim = Simulator(synthetic_trajectory(duration=30.0, pause=5.0), num_pts=100, num_pts_plane=100, sim_distance_threshold=-1.0)
    r = closed_loop.run_session(sim, n_frames=120, C=11, max_slam=25, zupt={})
    z = r["zupt_frames"]
    n_still = int(z.sum())
    assert 30 <= n_still <= 45 and z[:n_still].all() and not z[n_still:].any(), np.nonzero(z)[0]
    assert r["e_pos"][:n_still].max() < 5e-3 and np.degrees(r["e_ori"][:n_still]).max() < 0.05
    assert (r["counts"][:n_still, :4] == 0).all()                      # no update, no clone while standing
    assert r["counts"][n_still + 15:, 1].min() >= 10                   # landmarks are tracked once it moves
    assert r["rmse_pos"] < 0.3 and r["e_pos"].max() < 0.5, (r["rmse_pos"], r["e_pos"].max())


def test_filter_session_rejects_inconsistent_bookkeeping(hiplib):
    """The session does not guess: a SLAM measurement for a landmark that is not in the state, or a window slot outside the
    C + 1 clones, is an error of the caller's tracker-side bookkeeping (no silent drop, no crash)."""
    from ov_plane_amd.build import build_host

    build_host()
    from ov_plane_amd import closed_loop, hostlib
    from ov_plane_amd.sim import Simulator, synthetic_trajectory
    from ov_plane_amd.synth import PROP_OPTS

    C = 6
    sim = Simulator(synthetic_trajectory(duration=12.0), num_pts=30, num_pts_plane=30)
    imu, frames, _ = closed_loop.collect(sim, C + 3)
    init = closed_loop.initial_state(sim, frames, C)
    uv = np.full((1, C + 1, 2), 100.0, dtype=np.float32)
    nm = np.array([1], dtype=np.int32)
    for kind, slot0, msg in ((1, C, "-22"), (0, C + 1, "-21")):
        ses = hostlib.Session(init, dict(PROP_OPTS), max_slam=5)
        ses.feed_imu(imu)
        slot = -np.ones((1, C + 1), dtype=np.int32)
        slot[0, 0] = slot0
        with pytest.raises(RuntimeError, match=msg):
            ses.step(frames[C + 1][0], uv, uv * 0, slot, nm, np.array([99999]), np.array([kind], dtype=np.int32))
        # the rejected call left the filter untouched: the same frame can be stepped again (inputs are validated before the
        # propagation; a second propagation to the same time would be fatal)
        out = ses.step(frames[C + 1][0], uv[:0], uv[:0], slot[:0], nm[:0], np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int32))
        assert np.isfinite(out["x16"]).all()
        ses.close()


def test_filter_session_marginalises_planes_nobody_observes(hiplib):
    """core/VioManager.cpp:513-534 -> StateHelper::merge_planes_and_marginalize every frame: with the tracker's plane list handed
    over, planes stay in the state while their features are tracked and leave it (three covariance columns each) as soon as
    the tracker reports them no longer - here from frame 25 on, by the test hook of closed_loop.run_session."""
    from ov_plane_amd.build import build_host

    build_host()
    from ov_plane_amd import closed_loop
    from ov_plane_amd.sim import Simulator, synthetic_trajectory

    sim = Simulator(synthetic_trajectory(duration=20.0), num_pts=60, num_pts_plane=120)
    r = closed_loop.run_session(sim, n_frames=40, C=8, planes=2, forget_planes_at=25)
    n_planes = r["counts"][:, 5]
    assert n_planes[:25].max() >= 1, n_planes
    assert (n_planes[25:] == 0).all(), n_planes
    assert np.isfinite(r["traj"]).all() and r["rmse_pos"] < 0.3


def _plane_gate_tool():
    import importlib.util

    spec = importlib.util.spec_from_file_location(
        "plane_gate_agreement", os.path.join(os.path.dirname(GOLD), "..", "tools", "plane_gate_agreement.py"))
    pga = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pga)
    return pga


GATE_DECIDED = 3.0   # an ensemble "has decided" a plane when all four builds sit on the same side of the threshold by at least this:
                     # there the device must decide the same, without exception (measured: the closest build of a plane on which the
                     # device sits on the other side of a unanimous ensemble is 1.71 from the threshold; 1470 planes)
GATE_CLOSE = 1.5     # ... and between 1.5 and 3.0 it may not, rarely (a rate is bounded, not every plane)
GATE_BAND = 30.4     # largest distance between two builds of the oracle on one plane of the fixture (1470 planes, four roundings);
                     # round 5 quoted 18.2 from two builds over 1120 planes - the same distribution sampled less often


def _ensemble_of(kw):
    """Rows of tests/golden/plane_gate_ensemble.npz for one scene: dict(ok, dof, thr, chi2 [planes, 4 builds], decided, decision).
    decided: every build on the same side of the threshold by GATE_DECIDED or more."""
    import json

    fx = _plane_gate_tool().load_fixture()
    want = json.dumps(kw, sort_keys=True)
    hit = [s for s, k in enumerate(fx["scenes"]) if json.dumps(k, sort_keys=True) == want]
    assert hit, "scene not in the fixture (tests/golden/make_plane_gate_ensemble.py): %s" % want
    rows = np.where(fx["scene"] == hit[0])[0]
    E, thr = fx["chi2"][rows], fx["thr"][rows]
    side = E <= thr[:, None]
    decided = (side.all(axis=1) | (~side).all(axis=1)) & (np.abs(E - thr[:, None]).min(axis=1) >= GATE_DECIDED)
    return dict(ok=fx["ok"][rows], dof=fx["dof"][rows], thr=thr, chi2=E, decided=decided, decision=side[:, 0])


def test_plane_gate_against_the_oracle_ensemble(hiplib, oracle):
    """Plane-level chi2 gate (update/UpdaterMSCKF.cpp:607-631 on the system update/UpdaterPlane.cpp:545-551 truncates) at
    chi2_multipler = 1 - the value of the real-data configs (config/euroc_mav/estimator_config.yaml:155) - on the 63 scenes of
    tests/golden/plane_gate_ensemble.npz: 50 of config 3's shape, configs 3 and 4 at five seeds each, the frames of the whole-step
    tests (1470 planes).

    The reference's statistic carries (kept rows - rank) rows of a rank-deficient Givens sweep whose content is decided by rounding:
    FOUR builds of the oracle (plain, fma, x87, re-associated; oracle/Makefile) differ from each other by up to 25 on one plane and
    on 4 % of the decisions.  One build is therefore one sample of what the reference answers, and the device - which computes the
    deterministic part plus 0.96 x the expected energy of those rows (k_chol2.hip, OVP_PLANE_NOISE_KAPPA) - is held to the ENSEMBLE.
    All five run on the plain build's accept / reject sequence, so they see the same state at every plane.  Contract:
      (1) same row count in the test (dof) on every plane;
      (2) |chi2_device - chi2_build| <= the builds' own largest distance from each other (GATE_BAND), for every plane and build;
      (3) no bias against the ensemble mean: |mean| <= 0.2 over all planes, <= 0.3 for in-state / out-of-state planes;
      (4) the device is closer to the centre of the ensemble than a build is (spread against the mean of the builds);
      (5) its decisions flip against a build LESS often than two builds flip against each other;
      (6) wherever the ensemble has decided (all builds on one side by >= 3.0, 89 % of the planes) the device decides the same - no
          exception; where all builds are on one side by >= 1.5 it does so on all but <= 0.2 % of the planes (measured: 1 of 1333);
      (7) unanimous planes on which the device sits on the other side: <= 0.6 % (measured 5 of 1374), the device within 4.0 of the
          threshold on each - a deterministic statistic cannot do better against samples that scatter by +-4.4 around their centre
          (the device itself scatters by 2.3 around it); emulating ONE build's rounding would, and would be wrong for every other.
    Numbers of the committed run: profiles/r06_plane_gate_agreement.json."""
    pga = _plane_gate_tool()
    fx = pga.load_fixture()
    chi2_dev, dof_dev = pga.device_statistics(hiplib, fx)
    s = pga.analyse(fx, chi2_dev, dof_dev)
    assert s["scenes"] == 63 and s["planes"] >= 1400
    assert 0.7 < s["oracle_accept_rate"] < 0.95                       # the gate is really deciding at this multiplier
    assert s["dof_mismatch"] == 0                                     # (1)
    band = s["interbuild_band"]
    assert 18.0 < band <= GATE_BAND                                   # the reference's own statistic moves this far between builds
    for nm, v in s["device_vs_build"].items():                        # (2)
        assert v["abs_max"] <= band, (nm, v)
    em = s["device_vs_ensemble_mean"]                                 # (3)  (standard error over 1470 planes: 0.06)
    assert abs(em["all"]["mean"]) <= 0.2 and abs(em["in_state"]["mean"]) <= 0.3 and abs(em["out_of_state"]["mean"]) <= 0.3, em
    E = fx["chi2"]                                                    # (4)
    build_spread = min(float((E[:, a] - np.delete(E, a, axis=1).mean(axis=1)).std()) for a in range(E.shape[1]))
    assert em["all"]["std"] < 0.75 * build_spread, (em["all"]["std"], build_spread)
    assert s["device_vs_build_flip_rate_mean"] < s["oracle_vs_oracle_flip_rate_mean"], s   # (5)
    thr = fx["thr"]
    side = E <= thr[:, None]
    one_side = side.all(axis=1) | (~side).all(axis=1)
    nearest = np.abs(E - thr[:, None]).min(axis=1)
    decided = one_side & (nearest >= GATE_DECIDED)
    close = one_side & (nearest >= GATE_CLOSE)
    dev_side = chi2_dev <= thr
    assert decided.mean() > 0.85                                      # most planes are decided ones
    assert (dev_side[decided] == side[decided, 0]).all()              # (6) no exception
    assert (dev_side[close] != side[close, 0]).mean() <= 0.002, int((dev_side[close] != side[close, 0]).sum())
    v = s["unanimous_violations"]                                     # (7)
    assert len(v) <= 0.006 * s["ensemble_unanimous"], v
    assert all(x["device_margin"] < 4.0 and x["nearest_build_margin"] < GATE_DECIDED for x in v), v
    # the fixture is this oracle's output: the plain build, run here, reproduces its column on a config-3 frame bit for bit or nearly so
    kw = dict(C=30, F=2000, seed=11, n_planes=20, feats_per_plane=50, planes_in_state_frac=0.5, chi2_mult=1.0)
    live = oracle.msckf_plane_update(make_scene(**kw))
    ens = _ensemble_of(kw)
    assert (live["plane_ok"] == ens["ok"]).all()
    assert np.abs(live["plane_chi2"] - ens["chi2"][:, 0]).max() <= 1e-6 * np.abs(ens["chi2"][:, 0]).max()


@pytest.mark.parametrize("n", [5, 16, 31, 100, 197, 240, 256, 285, 287])
def test_tile_cholesky_with_border_row_matches_numpy(hiplib, n):
    """k_chol2 (the factorization every plane of the plane loop runs) on its own, through ovp_debug_chol2: factor of A + I, the
    border row z = L^-1 b that rides along as one more matrix row, the back substitution y = L^-T z on the register-resident
    factor, and the pivots - against numpy on a random SPD matrix.  Sizes cover one tile, partial last tiles, a border row that
    opens a tile row of its own (n a multiple of 16), the 17-panel-tile second elimination pass (n + 1 > 272) and the scratch-
    backed register configuration (n = 285, the state of BASELINE config 4)."""
    rng = np.random.default_rng(n)
    M = rng.standard_normal((n, n + 5))
    A = M @ M.T / n + 0.1 * np.eye(n)
    b = rng.standard_normal(n)
    ctx = hiplib.Context(288, 30, 8)
    out = ctx.debug_chol2(A, b, add_identity=True)
    assert out["rc"] == 0
    L = np.linalg.cholesky(A + np.eye(n))
    z = np.linalg.solve(L, b)
    y = np.linalg.solve(L.T, z)
    assert np.abs(out["L"][:n, :n] - L).max() < 1e-13
    assert np.abs(out["L"][n, :n] - z).max() < 1e-12 and np.abs(out["z"] - z).max() < 1e-12
    assert np.abs(out["y"] - y).max() < 1e-12
    assert np.abs(out["piv"] - np.diag(L) ** 2).max() < 1e-12
    # without a border, and a matrix that is not positive definite is reported, not factorized into garbage silently
    out2 = ctx.debug_chol2(A, None, add_identity=False)
    assert out2["rc"] == 0 and np.abs(out2["L"] - np.linalg.cholesky(A)).max() < 1e-13
    Abad = A.copy()
    Abad[n // 2, n // 2] = -1.0
    assert ctx.debug_chol2(Abad, None)["rc"] == -3  # OVP_E_NOTSPD
    ctx.close()


@pytest.mark.parametrize("n", [40, 240, 285])
def test_tile_cholesky_bad_and_dropped_pivots_terminate(hiplib, n):
    """The role hand-over of k_chol2 spins on LDS counters: a pivot that is not positive (NaNs from there on) or one that is dropped
    (pivot floor: the range part of the plane solve on a rank-deficient Gram) must not keep any wave from taking its steps.
    Positions: first column, inside the first tile column, first column of a later tile, the last tile column and the very last
    column - with and without the border row.  Not positive -> OVP_E_NOTSPD (and the call returns); dropped -> the factor of the
    matrix with that direction removed, zero entry in z, against a numpy elimination with the same rule."""
    rng = np.random.default_rng(1000 + n)
    ctx = hiplib.Context(288, 30, 8)
    M = rng.standard_normal((n, n + 7))
    A = M @ M.T / n + 0.05 * np.eye(n)
    b = rng.standard_normal(n)
    for pos in sorted({0, 5, 16, 16 * ((n - 1) // 16), n - 3, n - 1}):
        Abad = A.copy()
        Abad[pos, pos] = -2.0
        assert ctx.debug_chol2(Abad, None)["rc"] == -3, pos
        assert ctx.debug_chol2(Abad, b)["rc"] == -3, pos
    # rank-deficient positive semi-definite matrix: the directions completed at columns `null` are dropped
    null = sorted({0, 7, 16, 16 * ((n - 1) // 16) + 1, n - 2, n - 1})
    keep = [i for i in range(n) if i not in null]
    B = rng.standard_normal((n + 3, n))
    for c in null:  # column c = combination of the columns before it (column 0: zero)
        B[:, c] = B[:, :c] @ rng.standard_normal(c) / max(c, 1) if c else 0.0
    G = B.T @ B
    d = np.sqrt(np.where(np.diag(G) > 0, np.diag(G), 1.0))
    Gn = G / np.outer(d, d) + 1e-12 * np.eye(n)
    bn = (B.T @ rng.standard_normal(n + 3)) / d

    def chol_drop(Amat, rhs, floor):
        Aw, bw = Amat.copy(), rhs.copy()
        L, z, piv = np.zeros_like(Amat), np.zeros(len(rhs)), np.zeros(len(rhs))
        for k in range(len(rhs)):
            piv[k] = Aw[k, k]
            if not (piv[k] >= floor):
                continue
            l = Aw[k:, k] / np.sqrt(piv[k])
            L[k:, k] = l
            Aw[k:, k:] -= np.outer(l, l)
            z[k] = bw[k] / np.sqrt(piv[k])
            bw[k:] -= l * z[k]
        return L, z, piv

    Lr, zr, pr = chol_drop(Gn, bn, 1e-5)
    out = ctx.debug_chol2(Gn, bn, piv_floor=1e-5)
    assert out["rc"] == 0
    assert set(np.where(out["piv"] < 1e-5)[0]) == set(null) == set(np.where(pr < 1e-5)[0])
    assert np.abs(out["piv"][keep] - pr[keep]).max() < 1e-10
    assert np.abs(out["z"] - zr).max() < 1e-9 and np.abs(out["z"][null]).max() == 0.0
    assert np.abs(out["L"][:n, :n][np.ix_(keep, keep)] - Lr[np.ix_(keep, keep)]).max() < 1e-9
    ctx.close()


def test_point_update_may_always_ask_for_the_plane_mask(hiplib, oracle):
    """ovp_update_opts::skip_plane_used is valid behind EVERY ovp_msckf_plane_update of the batch: a frame without planes (nothing
    consumed) and the flag left set in the options handed to
    the plane loop itself (ignored there)."""
    sc = make_scene(C=8, F=60, seed=4, n_planes=2, feats_per_plane=12, chi2_mult=99999.0)
    ctx = hiplib.Context(sc.N, sc.C, sc.F)
    o = hiplib.opts_from_scene(sc)
    o.skip_plane_used = 1
    # (i) no planes in the batch
    ctx.cov_upload(sc.P)
    ctx.state_upload(sc)
    ctx.batch_upload_scene(sc)
    none = ctx.plane_update(o, sc.plane_id, np.zeros((0, 3)), np.zeros((0, 3)), np.zeros(0, dtype=np.int32))
    assert none["rc"] == 0 and not none["used"].any()
    pt = ctx.msckf_update(o)
    assert pt["accepted"].sum() > 0.8 * sc.F
    # (ii) the flag set in the plane loop's own options; the point update then skips exactly the consumed features
    ctx.cov_upload(sc.P)
    ctx.state_upload(sc)
    ctx.batch_upload_scene(sc)
    pl = ctx.plane_update(o, sc.plane_id, sc.cp, sc.cp_fej, sc.plane_state_id)
    assert pl["ok"].all() and pl["used"].sum() == 24
    pt = ctx.msckf_update(o)
    assert not pt["accepted"][pl["used"]].any() and pt["accepted"][~pl["used"]].sum() > 0.8 * (sc.F - 24)
    ctx.close()


def test_index_range_shards_of_the_leftovers_sum_to_the_unsharded_pair(hiplib):
    """Multi-GPU split of SURVEY 8(e) on one device: the frame is uploaded once, the plane loop runs on all of it, and each rank's
    share of the leftovers is an index range (ovp_batch_set_range via dist.leftover_range) with the consumed features masked on the
    device.  The information pairs of three such shards sum to the pair of the unsharded point update, the accept decisions are
    the same feature by feature, an empty shard contributes nothing."""
    from ov_plane_amd.dist import leftover_range

    sc = make_scene(C=10, F=150, seed=72, n_planes=4, feats_per_plane=20, chi2_mult=99999.0)
    ctx = hiplib.Context(sc.N, sc.C, sc.F)
    o = hiplib.opts_from_scene(sc)
    ld = ((sc.N + 15) // 16) * 16

    def after_planes():
        ctx.cov_upload(sc.P)
        ctx.state_upload(sc)
        ctx.batch_upload_scene(sc)
        return ctx.plane_update(o, sc.plane_id, sc.cp, sc.cp_fej, sc.plane_state_id)

    pl = after_planes()
    assert pl["used"].sum() == 80
    op = hiplib.opts_from_scene(sc)
    op.chi2_multiplier = 1.0
    op.skip_plane_used = 1
    ctx.build_gate_gram_async(op)
    ctx.sync()
    Ab_all = ctx.debug_read("Ab", (sc.N + 1, ld)).copy()
    full = ctx.msckf_update(op)
    Ab_sum = np.zeros_like(Ab_all)
    acc = np.zeros(sc.F, dtype=bool)
    seen = []
    world = 3
    for rank in range(world):
        pl_r = after_planes()
        assert (pl_r["used"] == pl["used"]).all()
        lo, hi, mine = leftover_range(pl_r["used"], rank, world)
        ctx.batch_set_range(lo, hi)
        ctx.build_gate_gram_async(op)
        ctx.sync()
        Ab_sum += ctx.debug_read("Ab", (sc.N + 1, ld))
        ctx.ekf_update_from_gram_async()
        r = ctx.fetch_results()
        assert not r["accepted"][np.setdiff1d(np.arange(sc.F), mine)].any()
        acc |= r["accepted"]
        seen.append(mine)
    assert (np.sort(np.concatenate(seen)) == np.where(~pl["used"])[0]).all()
    assert (acc == full["accepted"]).all()
    scale = np.abs(Ab_all).max()
    assert np.abs(Ab_sum - Ab_all).max() < 1e-11 * scale
    # an empty shard (more ranks than leftovers would give one): zero pair, nothing accepted
    after_planes()
    ctx.batch_set_range(5, 5)
    ctx.build_gate_gram_async(op)
    ctx.sync()
    assert np.abs(ctx.debug_read("Ab", (sc.N + 1, ld))).max() == 0.0
    ctx.ekf_update_from_gram_async()
    assert not ctx.fetch_results()["accepted"].any()
    ctx.close()


def test_plane_loop_with_every_plane_rejected_or_absent(hiplib):
    """Edge cases of the plane loop: (i) every plane fails its chi2 test (multiplier 1e-9) - the state tables, the used mask and
    the covariance stay what they were (the covariance is re-materialised from its factor: equal to rounding); (ii) the same through
    force_decision = 0; (iii) a plane batch in which no feature lies on a plane - nothing is enqueued at all."""
    sc = make_scene(C=9, F=90, seed=17, n_planes=3, feats_per_plane=15, chi2_mult=1e-9)
    ctx = hiplib.Context(sc.N, sc.C, sc.F)
    o = hiplib.opts_from_scene(sc)
    d = np.sqrt(np.diag(sc.P))
    for force in (None, np.zeros(3, dtype=np.uint8)):
        ctx.cov_upload(sc.P)
        ctx.state_upload(sc)
        ctx.batch_upload_scene(sc)
        if force is not None:
            o.chi2_multiplier = 99999.0
        out = ctx.plane_update(o, sc.plane_id, sc.cp, sc.cp_fej, sc.plane_state_id, force_decision=force)
        assert not out["ok"].any() and not out["used"].any() and (out["dof"] > 0).all() and (out["chi2"] > 0).all()
        assert np.abs(out["dx"]).max() == 0.0
        assert (np.abs(ctx.cov_download() - sc.P) / np.outer(d, d)).max() < 1e-12
        # the point update that follows sees every feature (nothing was consumed)
        o2 = hiplib.opts_from_scene(sc)
        o2.chi2_multiplier = 1.0
        o2.skip_plane_used = 1
        assert ctx.msckf_update(o2)["accepted"].sum() > 0.8 * sc.F
    ctx.cov_upload(sc.P)
    ctx.batch_upload_scene(sc)
    none = ctx.plane_update(o, np.zeros(sc.F, dtype=np.int32), sc.cp, sc.cp_fej, sc.plane_state_id)
    assert not none["ok"].any() and (none["dof"] == 0).all() and np.array_equal(ctx.cov_download(), sc.P)
    ctx.close()


def test_cov_initialize_one_call_matches_separate_calls_and_rejects_cleanly(hiplib):
    """ovp_cov_initialize (gate + initialize_invertible + EKFUpdate, state/StateHelper.cpp:448-487) against the same three steps
    in numpy; a rejected candidate leaves the covariance and its dimension bit-identical; problems outside the entry's limits
    are refused with OVP_E_CAPACITY before anything is touched."""
    rng = np.random.default_rng(3)
    sc = make_scene(C=11, F=4, seed=12)
    n = sc.P.shape[0]
    cols, k, rup, r = 24, 3, 17, 0.7
    ids = np.sort(rng.choice(n, cols, replace=False)).astype(np.int32)
    Hx = rng.standard_normal((k, cols)) * 5.0
    Hu = rng.standard_normal((rup, cols)) * 5.0
    HL = rng.standard_normal((k, k)) + 3.0 * np.eye(k)
    HLi = np.linalg.inv(HL)
    Ri = r * np.eye(k)
    res = rng.standard_normal(rup)
    # numpy: gate, augmentation (StateHelper.cpp:531-566), update (:159-187)
    Pm = sc.P[np.ix_(ids, ids)]
    S = Hu @ Pm @ Hu.T + r * np.eye(rup)
    chi2_ref = float(res @ np.linalg.solve(S, res))
    M = sc.P[:, ids] @ Hx.T
    Pll = HLi @ (Hx @ Pm @ Hx.T + Ri) @ HLi.T
    Pxl = -M @ HLi.T
    P2 = np.zeros((n + k, n + k))
    P2[:n, :n] = sc.P
    P2[:n, n:] = Pxl
    P2[n:, :n] = Pxl.T
    P2[n:, n:] = Pll
    Hf = np.zeros((rup, n + k))
    Hf[:, ids] = Hu
    Sf = Hf @ P2 @ Hf.T + r * np.eye(rup)
    K = P2 @ Hf.T @ np.linalg.inv(Sf)
    dx_ref = K @ res
    P3 = P2 - K @ Hf @ P2
    ctx = hiplib.Context(n + 8, sc.C, sc.F)
    ctx.cov_upload(sc.P)
    # reject: nothing changes
    ok, chi2, _ = ctx.cov_initialize(Hx, Hu, ids, HLi, Ri, res, r, chi2_ref * 0.5)
    assert not ok and abs(chi2 - chi2_ref) < 1e-9 * max(1.0, chi2_ref)
    assert ctx.cov_size() == n and np.array_equal(ctx.cov_download(), sc.P)
    # outside the limits: refused
    with pytest.raises(hiplib.OvpError) as e:
        ctx.cov_initialize(Hx, rng.standard_normal((81, cols)), ids, HLi, Ri, rng.standard_normal(81), r, 1e9)
    assert e.value.code == hiplib.OVP_E_CAPACITY and ctx.cov_size() == n
    # accept
    ok, chi2, dx = ctx.cov_initialize(Hx, Hu, ids, HLi, Ri, res, r, chi2_ref * 2.0)
    assert ok and ctx.cov_size() == n + k
    assert relP(ctx.cov_download(), P3) < TOL_P
    assert np.abs(dx - dx_ref).max() < TOL_DX
    # accept without the update rows being applied (do_update = false), and with no update rows at all
    ctx.cov_upload(sc.P)
    ok, _, dx = ctx.cov_initialize(Hx, Hu, ids, HLi, Ri, res, r, 1e9, do_update=False)
    assert ok and relP(ctx.cov_download(), P2) < TOL_P and not dx.any()
    ctx.cov_upload(sc.P)
    ok, chi2, dx = ctx.cov_initialize(Hx, None, ids, HLi, Ri, None, 1.0, 0.0)
    assert ok and chi2 == 0.0 and relP(ctx.cov_download(), P2) < TOL_P
    ctx.close()


def test_plane_solve_on_two_workgroups_equals_the_one_workgroup_solve(hiplib, oracle):
    """k_chol2's update part split by tile columns over two workgroups (exported panels, imported by the second one; gate in the
    second, commit in the first; default from 17 tile columns on, i.e. BASELINE config 4) forced at N = 240 with two split points,
    against the single-workgroup solve: same decisions at multiplier 1, corrections and covariance equal to rounding, and both
    match the oracle."""
    sc = make_scene(C=30, F=360, seed=21, n_planes=6, feats_per_plane=40, planes_in_state_frac=0.5, chi2_mult=1.0)
    assert sc.N == 219
    ref = oracle.msckf_plane_update(sc)
    outs = []
    old = os.environ.get("OVP_C2_SPLIT")
    try:
        for h in ("0", "5", "8"):
            os.environ["OVP_C2_SPLIT"] = h
            ctx = hiplib.Context(sc.N, sc.C, sc.F)
            ctx.cov_upload(sc.P)
            ctx.state_upload(sc)
            ctx.batch_upload_scene(sc)
            out = ctx.plane_update(hiplib.opts_from_scene(sc), sc.plane_id, sc.cp, sc.cp_fej, sc.plane_state_id,
                                   force_decision=ref["plane_ok"].astype(np.uint8))
            out["P"] = ctx.cov_download()
            outs.append(out)
            ctx.close()
    finally:
        if old is None:
            os.environ.pop("OVP_C2_SPLIT", None)
        else:
            os.environ["OVP_C2_SPLIT"] = old
    base = outs[0]
    assert relP(base["P"], ref["P"]) < TOL_P and ref["plane_ok"].sum() >= 3
    for o in outs[1:]:
        assert (o["ok"] == base["ok"]).all() and (o["used"] == base["used"]).all()
        # (the statistic of a later plane sees the corrections of the earlier ones: a difference of 1e-12 in dx - rounding, the two
        # solves add in different orders - moves a chi2 of ~200 by ~1e-6 through residuals of hundreds of pixels per unit state)
        assert np.abs(o["chi2"] - base["chi2"]).max() < 1e-7 * np.abs(base["chi2"]).max()
        assert np.abs(o["dx"] - base["dx"]).max() < 1e-11
        assert relP(o["P"], base["P"]) < 1e-11


def test_plane_loop_is_bitwise_repeatable(hiplib):
    """The wave roles of k_chol2 hand data over through LDS counters (factorization, back substitution): forty runs of the
    config-3 plane loop on the same frame must give the same bits (dx of every plane, decisions, covariance)."""
    sc = make_scene(C=30, F=2000, seed=0, n_planes=20, feats_per_plane=50, planes_in_state_frac=0.5, chi2_mult=1.0)
    ctx = hiplib.Context(sc.N, sc.C, sc.F)
    o = hiplib.opts_from_scene(sc)
    ref = None
    for _ in range(40):
        ctx.cov_upload(sc.P)
        ctx.state_upload(sc)
        ctx.batch_upload_scene(sc)
        pl = ctx.plane_update(o, sc.plane_id, sc.cp, sc.cp_fej, sc.plane_state_id)
        key = (pl["dx"].tobytes(), pl["ok"].tobytes(), ctx.cov_download().tobytes())
        if ref is None:
            ref = key
            assert pl["ok"].sum() >= 10
        assert key == ref
    ctx.close()


def test_few_row_updates_on_a_singular_covariance_and_at_the_row_limit(hiplib):
    """The S-form kernels (csrc/k_init.hip) need no factor of P: a dense update and a landmark initialisation right after an exact
    stochastic clone (P positive SEMI-definite, ovp_cov_clone copies rows and columns) against the reference form in numpy; the
    row limit of the path (80 rows) and the first size behind it (information form) give the same update."""
    rng = np.random.default_rng(21)
    sc = make_scene(C=6, F=4, seed=5)
    n = sc.P.shape[0]
    ctx = hiplib.Context(n + 16, sc.C + 2, sc.F)
    ctx.cov_upload(sc.P)
    src = int(sc.ids["clones"][0])
    ctx.cov_clone(src, 6)                      # exact copy: singular covariance of dimension n + 6
    Pc = ctx.cov_download()
    assert np.linalg.matrix_rank(Pc) == n and np.abs(Pc[n:, n:] - Pc[src:src + 6, src:src + 6]).max() < 1e-12 * np.abs(Pc).max()
    N = n + 6
    cols = np.concatenate([np.arange(src, src + 6), np.arange(n, n + 6), np.arange(3)]).astype(np.int32)
    H = rng.standard_normal((12, len(cols))) * 10.0
    res = rng.standard_normal(12)
    Hf = np.zeros((12, N))
    Hf[:, cols] = H
    S = Hf @ Pc @ Hf.T + np.eye(12)
    K = Pc @ Hf.T @ np.linalg.inv(S)
    dx, _ = ctx.ekf_update(H, cols, res)
    assert np.abs(dx - K @ res).max() < TOL_DX
    P1 = Pc - K @ Hf @ Pc
    assert relP(ctx.cov_download(), P1) < TOL_P
    # landmark initialisation on the (still singular) result
    k, rup = 3, 9
    Hx = rng.standard_normal((k, len(cols))) * 5.0
    Hu = rng.standard_normal((rup, len(cols))) * 5.0
    HLi = np.linalg.inv(rng.standard_normal((k, k)) + 3.0 * np.eye(k))
    r = rng.standard_normal(rup)
    ok, chi2, dx2 = ctx.cov_initialize(Hx, Hu, cols, HLi, np.eye(k), r, 1.0, 1e9)
    Pm = P1[np.ix_(cols, cols)]
    M = P1[:, cols] @ Hx.T
    P2 = np.zeros((N + k, N + k))
    P2[:N, :N] = P1
    P2[:N, N:] = -M @ HLi.T
    P2[N:, :N] = P2[:N, N:].T
    P2[N:, N:] = HLi @ (Hx @ Pm @ Hx.T + np.eye(k)) @ HLi.T
    Hf2 = np.zeros((rup, N + k))
    Hf2[:, cols] = Hu
    S2 = Hf2 @ P2 @ Hf2.T + np.eye(rup)
    K2 = P2 @ Hf2.T @ np.linalg.inv(S2)
    assert ok and abs(chi2 - r @ np.linalg.solve(Hu @ Pm @ Hu.T + np.eye(rup), r)) < 1e-8 * max(1.0, chi2)
    assert np.abs(dx2 - K2 @ r).max() < TOL_DX and relP(ctx.cov_download(), P2 - K2 @ Hf2 @ P2) < TOL_P
    ctx.close()
    # 80 rows (S-form) and 81 rows (information form) of the same random system agree with numpy
    sc = make_scene(C=11, F=4, seed=6)
    n = sc.P.shape[0]
    cols = np.arange(n, dtype=np.int32)[:60]
    for rows in (80, 81):
        ctx = hiplib.Context(n, sc.C, sc.F)
        ctx.cov_upload(sc.P)
        H = rng.standard_normal((rows, len(cols))) * 3.0
        res = rng.standard_normal(rows)
        Hf = np.zeros((rows, n))
        Hf[:, cols] = H
        K = sc.P @ Hf.T @ np.linalg.inv(Hf @ sc.P @ Hf.T + np.eye(rows))
        dx, _ = ctx.ekf_update(H, cols, res)
        assert np.abs(dx - K @ res).max() < TOL_DX
        assert relP(ctx.cov_download(), sc.P - K @ Hf @ sc.P) < TOL_P
        ctx.close()


def _slam_plane_args(sc, use_planes):
    if not use_planes:
        return None, None, None
    pid = np.asarray(sc.plane_id, dtype=np.int64)
    sid = np.where(pid > 0, np.asarray(sc.plane_state_id)[np.maximum(pid, 1) - 1], -1).astype(np.int32)
    return sid, np.asarray(sc.cp)[np.maximum(pid, 1) - 1], np.asarray(sc.cp_fej)[np.maximum(pid, 1) - 1]


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [
    dict(C=11, n_slam=12, seed=3, outliers=2),
    dict(C=11, n_slam=14, seed=4, n_planes=3, outliers=2, wrong_plane=3),   # plane rows + the no-plane fallback
    dict(C=6, n_slam=5, seed=5, do_fej=False),
    dict(C=8, n_slam=8, seed=6, fisheye=True),
    dict(C=5, n_slam=20, seed=8, ragged=False, outliers=1),                 # 200 stacked rows: information form behind the gate
    dict(C=4, n_slam=9, seed=9, ragged=False, n_planes=2, wrong_plane=2),   # <= 80 stacked rows with plane rows: S-form
    dict(C=30, n_slam=6, seed=10, ragged=False, n_planes=2, wrong_plane=1, outliers=1),  # 90-row blocks (full-length tracks)
])
def test_slam_update_on_the_device_matches_oracle(hiplib, oracle, kw):
    """ovp_slam_update (csrc/k_slam.hip) = UpdaterSLAM::update (update/UpdaterSLAM.cpp:424-673) without the covariance leaving the
    device: rows of a landmark of the state, chi2 against the resident P, the no-plane fallback, stacking, EKFUpdate - statuses,
    statistics, correction and covariance against ovo_slam_update."""
    from ov_plane_amd.synth import make_slam_scene

    capi = hiplib
    sc = make_slam_scene(**kw)
    use_planes = kw.get("n_planes", 0) > 0
    ref = oracle.slam_update(sc, sc.lm_id, use_planes=use_planes)
    assert ref["rc"] == 0
    ctx = capi.Context(sc.N, sc.C, sc.F, device=0)
    ctx.cov_upload(sc.P)
    ctx.state_upload(sc)
    sid, cp, cpf = _slam_plane_args(sc, use_planes)
    out = ctx.slam_update(capi.opts_from_scene(sc), sc.uv, sc.clone_idx, sc.n_meas, sc.p_FinG, sc.p_FinG_fej, sc.lm_id, sid, cp, cpf)
    assert ((out["status"] > 0) == ref["accepted"]).all()
    assert ((out["status"] == 2) == (ref["fellback"] & ref["accepted"])).all()
    if kw.get("wrong_plane", 0):
        assert (out["status"] == 2).any()
    if kw.get("outliers", 0):
        assert (out["status"] == 0).any()
    big = ref["chi2"] < 1e200
    assert np.abs(out["chi2"][big] - ref["chi2"][big]).max() <= 1e-8 * max(1.0, np.abs(ref["chi2"][big]).max())
    assert np.abs(out["dx"] - ref["dx"]).max() < TOL_DX
    assert relP(ctx.cov_download(), ref["P"]) < TOL_P
    assert out["info"].n_accepted == int(ref["accepted"].sum())
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("rep", [1, 2, 3, 4])
def test_slam_update_on_the_device_with_host_built_blocks(hiplib, oracle, rep):
    """Landmarks in an anchored / inverse-depth representation hand their dense block [H_x | H_f] to ovp_slam_update (pre_*), the
    gate against the resident covariance and the update are the device's: against ovo_slam_update_rep; one GLOBAL_3D landmark
    built on the device rides in the same call."""
    from ov_plane_amd.synth import make_slam_scene

    capi = hiplib
    sc = make_slam_scene(C=8, n_slam=8, seed=6, outliers=1)
    anchor = 2
    reps = np.full(sc.F, rep)
    reps[0] = 0
    ref = oracle.slam_update(sc, sc.lm_id, rep=reps, anchor=np.full(sc.F, anchor))
    assert ref["rc"] == 0 and not ref["accepted"].all()
    pre = [None]
    for f in range(1, sc.F):
        Hf, Hx, res, order = oracle.feature_jacobian_full_rep(sc, f, rep, anchor)
        ids = [i + k for (i, s) in order for k in range(s)] + [int(sc.lm_id[f]) + k for k in range(Hf.shape[1])]
        pre.append((np.hstack([Hx, Hf]), ids, res))
    ctx = capi.Context(sc.N, sc.C, sc.F, device=0)
    ctx.cov_upload(sc.P)
    ctx.state_upload(sc)
    out = ctx.slam_update(capi.opts_from_scene(sc), sc.uv, sc.clone_idx, sc.n_meas, sc.p_FinG, sc.p_FinG_fej, sc.lm_id, pre=pre)
    assert ((out["status"] > 0) == ref["accepted"]).all()
    assert np.abs(out["chi2"] - ref["chi2"]).max() <= 1e-8 * max(1.0, np.abs(ref["chi2"]).max())
    assert np.abs(out["dx"] - ref["dx"]).max() < TOL_DX
    assert relP(ctx.cov_download(), ref["P"]) < TOL_P
    ctx.close()


@pytest.mark.gpu
def test_slam_update_on_the_device_single_inverse_depth(hiplib, oracle):
    """ANCHORED_INVERSE_DEPTH_SINGLE landmarks (update/UpdaterSLAM.cpp:478-515): linearised as the MSCKF inverse depth, the depth
    column moved to the state side, the bearing projected out - here with a Householder basis of the left nullspace instead of the
    reference's Givens sweep (same row space: the statistic, the correction and the covariance do not depend on the basis).  A
    landmark with one observation is left out (required_meas = 2, :409-418)."""
    from ov_plane_amd.synth import make_slam_scene

    capi = hiplib
    sc = make_slam_scene(C=8, n_slam=8, seed=6, outliers=1)
    sc.n_meas[3] = 1
    anchor = 2
    ref = oracle.slam_update(sc, sc.lm_id, rep=np.full(sc.F, 5), anchor=np.full(sc.F, anchor))
    assert ref["rc"] == 0 and not ref["accepted"][3] and ref["accepted"].sum() >= 5 and not ref["accepted"][-1]
    keep = [f for f in range(sc.F) if sc.n_meas[f] >= 2]
    pre = []
    for f in keep:
        Hf, Hx, res, order = oracle.feature_jacobian_full_rep(sc, f, 4, anchor)
        Q, _ = np.linalg.qr(Hf[:, :2], mode="complete")
        N = Q[:, 2:]
        ids = [i + k for (i, s) in order for k in range(s)] + [int(sc.lm_id[f])]
        pre.append((N.T @ np.hstack([Hx, Hf[:, 2:3]]), ids, N.T @ res))
    ctx = capi.Context(sc.N, sc.C, sc.F, device=0)
    ctx.cov_upload(sc.P)
    ctx.state_upload(sc)
    out = ctx.slam_update(capi.opts_from_scene(sc), sc.uv[keep], sc.clone_idx[keep], sc.n_meas[keep], sc.p_FinG[keep],
                          sc.p_FinG_fej[keep], sc.lm_id[keep], pre=pre)
    assert ((out["status"] > 0) == ref["accepted"][keep]).all()
    assert np.abs(out["chi2"] - ref["chi2"][keep]).max() <= 1e-8 * max(1.0, np.abs(ref["chi2"]).max())
    assert np.abs(out["dx"] - ref["dx"]).max() < TOL_DX
    assert relP(ctx.cov_download(), ref["P"]) < TOL_P
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [3, 4])
def test_whole_step_under_the_devices_own_plane_decisions(hiplib, oracle, cfg):
    """BASELINE configs 3 and 4 with chi2_multipler = 1 on both levels and NOBODY imposing decisions on the device: its plane loop
    takes its own accept / reject sequence (five seeds each).  Two statements, for EVERY seed:
      (a) where that sequence leaves the plain oracle's, it does so at a plane the ensemble of four oracle builds has not decided
          (tests/golden/plane_gate_ensemble.npz: not all builds on one side of the threshold by 1.5) - up to there the sequences are
          equal;
      (b) on the sequence the device took, the reference algorithm gives the device's answer: the oracle re-run with the device's
          decisions imposed (ovo_set_plane_force) has the same state and covariance behind the loop, its statistic within GATE_BAND of
          the device's on every plane, and the point update on the leftovers - per-feature gate decisions, correction, covariance -
          agrees to the path's tolerances.
    Round 5 compared in full only the seeds whose sequence happened to equal the oracle's (3 of 5 at config 3, 1 of 5 at config 4)."""
    from ov_plane_amd.synth import Scene

    kw0 = (dict(F=2000, n_planes=20) if cfg == 3 else dict(F=8000, n_planes=50))
    n_equal = 0
    for seed in (11, 12, 13, 14, 15):
        kw = dict(C=30, seed=seed, feats_per_plane=50, planes_in_state_frac=0.5, chi2_mult=1.0, **kw0)
        sc = make_scene(**kw)
        ens = _ensemble_of(kw)
        ctx = hiplib.Context(sc.N, sc.C, sc.F)
        ctx.cov_upload(sc.P)
        ctx.state_upload(sc)
        ctx.batch_upload_scene(sc)
        o = hiplib.opts_from_scene(sc)
        pl = ctx.plane_update(o, sc.plane_id, sc.cp, sc.cp_fej, sc.plane_state_id)
        differ = np.where(pl["ok"] != ens["ok"])[0]
        if len(differ):                                                            # (a)
            k = int(differ[0])
            assert not ens["decided"][k], (seed, k, ens["chi2"][k], ens["thr"][k], pl["chi2"][k])
            ref_pl = oracle.msckf_plane_update(sc, force=pl["ok"])
        else:
            n_equal += 1
            ref_pl = oracle.msckf_plane_update(sc)
            assert (ref_pl["plane_ok"] == ens["ok"]).all()
        assert (ref_pl["plane_ok"] == pl["ok"]).all()                              # (b)
        assert (pl["used"] == ref_pl["used"]).all() and (pl["dof"] == ref_pl["plane_rows"]).all(), seed
        assert np.abs(pl["chi2"] - ref_pl["plane_chi2"]).max() <= GATE_BAND, seed
        cq, cpos, calq, calp, intr, cp = _apply_plane_dx(sc, pl["dx"], pl["ok"])
        assert np.abs(cpos - ref_pl["clone_p"]).max() < TOL_DX and np.abs(cq - ref_pl["clone_q"]).max() < TOL_DX, seed
        assert np.abs(intr - ref_pl["intr"]).max() < TOL_DX and np.abs(cp - ref_pl["cp"]).max() < TOL_DX, seed
        sc2 = Scene(sc)
        for key in ("P", "clone_q", "clone_p", "calib_q", "calib_p", "intr", "cp"):
            sc2[key] = ref_pl[key]
        rest = np.where(~ref_pl["used"])[0]
        ref = oracle.msckf_point_update_omp(sc2, feats=rest)
        o.skip_plane_used = 1
        out = ctx.msckf_update(o)
        P = ctx.cov_download()
        ctx.close()
        acc = np.asarray(out["accepted"]).astype(bool)
        assert (acc[rest] == ref["accepted"]).all() and not acc[ref_pl["used"]].any(), seed
        assert np.abs(out["dx"] - ref["dx"]).max() < TOL_DX and relP(P, ref["P"]) < TOL_P, seed
    assert n_equal >= 1, n_equal   # (some frame keeps the oracle's whole sequence: the gate is not flipping everywhere)


@pytest.mark.gpu
def test_native_rccl_sharded_update_on_one_rank_is_the_plain_update(hiplib, oracle):
    """ovp_msckf_update_sharded (SURVEY 8e from C: index range of the resident batch -> pair -> ncclAllReduce on the context's stream
    -> update) with a communicator of ONE rank created through the library's own RCCL binding (ovp_rccl_unique_id /
    ovp_rccl_comm_create): bit-equal to ovp_msckf_update, behind a plane loop (leftovers only) and without one; and two "ranks" played
    one after the other on this GPU - pairs summed by hand - reproduce the one-rank result (the split the 8-GPU run takes)."""
    capi = hiplib
    comm = capi.rccl_comm_create(capi.rccl_unique_id(), 0, 1, 0)
    try:
        for kw in (dict(C=12, F=300, seed=71, chi2_mult=1.0),
                   dict(C=12, F=260, seed=72, n_planes=5, feats_per_plane=30, planes_in_state_frac=0.6, chi2_mult=1.0)):
            sc = make_scene(**kw)
            planes = sc.cp.shape[0] > 0
            outs = []
            for sharded in (False, True):
                ctx = capi.Context(sc.N, sc.C, sc.F)
                ctx.cov_upload(sc.P)
                ctx.state_upload(sc)
                ctx.batch_upload_scene(sc)
                o = capi.opts_from_scene(sc)
                if planes:
                    pl = ctx.plane_update(o, sc.plane_id, sc.cp, sc.cp_fej, sc.plane_state_id)
                    assert pl["ok"].any()
                    o.skip_plane_used = 1
                out = ctx.msckf_update_sharded(o, comm, 0, 1) if sharded else ctx.msckf_update(o)
                out["P"] = ctx.cov_download()
                outs.append(out)
                if sharded:
                    rest = np.where(~pl["used"])[0] if planes else np.arange(sc.F)
                    assert out["shard"] == (int(rest[0]), int(rest[-1]) + 1)
                    if not planes:
                        # two ranks, one after the other: every rank's pair from its share, summed, is the pair of the whole batch
                        ld = ((sc.N + 15) // 16) * 16
                        cut = len(rest) // 2 + len(rest) % 2
                        Ab = []
                        for half in (rest[:cut], rest[cut:], rest):
                            ctx.cov_upload(sc.P)
                            ctx.batch_set_range(int(half[0]), int(half[-1]) + 1)
                            ctx.build_gate_gram_async(o)
                            ctx.sync()
                            Ab.append(ctx.debug_read("Ab", (sc.N + 1, ld)).copy())
                            ctx.ekf_update_from_gram_async()
                            ctx.fetch_results()
                        ctx.batch_set_range(-1, -1)
                        N = sc.N   # (the pair proper: the padding columns behind N are not part of it and are never written)
                        assert np.abs((Ab[0] + Ab[1] - Ab[2])[: N + 1, :N]).max() <= 1e-12 * np.abs(Ab[2][: N + 1, :N]).max()
                ctx.close()
            a, b = outs
            assert (a["accepted"] == b["accepted"]).all() and a["accepted"].sum() > 10
            assert np.array_equal(a["dx"], b["dx"]) and np.array_equal(a["P"], b["P"]) and np.array_equal(a["chi2"], b["chi2"])
    finally:
        capi.rccl_comm_destroy(comm)


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(C=11, F=90, seed=81, chi2_mult=1.0, ragged=True), dict(C=30, F=60, seed=82, chi2_mult=0.7)])
def test_dense_blocks_join_the_batch_in_one_update(hiplib, oracle, kw):
    """ovp_msckf_dense_blocks: features the batch format cannot carry (a track of more than OVP_MAX_MEAS observations, a second
    camera's observations) enter the update as dense blocks [H after the nullspace projection | columns | residual], are gated
    against the resident covariance like every feature (update/UpdaterMSCKF.cpp:739-757) and their information pair joins the
    batch's: ONE EKF update (:767-814).  Here a third of an ordinary scene's features is taken out of the batch and handed over as
    blocks (rows from the numpy restatement, nullspace by QR): gate decisions, correction and covariance must equal the update
    with every feature in the batch, and the oracle's."""
    from oracle import np_ref as R

    capi = hiplib
    sc = make_scene(**kw)
    ref = oracle.msckf_point_update(sc)
    dense = np.arange(sc.F) % 3 == 1
    blocks = []
    for f in np.where(dense)[0]:
        H_f, H_x, res, order = R.feature_jacobian_full(sc, int(f))
        Q, _ = np.linalg.qr(H_f, mode="complete")
        N = Q[:, H_f.shape[1]:]
        blocks.append((N.T @ H_x, R.order_cols(order), N.T @ res))
    ctx = capi.Context(sc.N, sc.C, sc.F)
    ctx.cov_upload(sc.P)
    ctx.state_upload(sc)
    acc_d, chi2_d = ctx.msckf_dense_blocks(sc.opts["chi2_mult"], blocks)
    assert (acc_d == ref["accepted"][dense]).all() and (kw["chi2_mult"] >= 1.0 or not acc_d.all())
    assert np.abs(chi2_d - ref["chi2"][dense]).max() <= 1e-8 * np.abs(ref["chi2"]).max()
    ctx.batch_upload_scene(sc, np.where(~dense)[0])
    out = ctx.msckf_update(capi.opts_from_scene(sc))
    assert (out["accepted"] == ref["accepted"][~dense]).all()
    assert np.abs(out["dx"] - ref["dx"]).max() < TOL_DX and relP(ctx.cov_download(), ref["P"]) < TOL_P
    # the sharded entry carries the pending pair as well (rank 0 brings it to the all-reduce, the others contribute zeros for it):
    # on a one-rank communicator the result is the same, bit for bit
    P_plain, dx_plain = ctx.cov_download(), out["dx"].copy()
    comm = capi.rccl_comm_create(capi.rccl_unique_id(), 0, 1, 0)
    try:
        ctx.cov_upload(sc.P)
        ctx.msckf_dense_blocks(sc.opts["chi2_mult"], blocks)
        ctx.batch_upload_scene(sc, np.where(~dense)[0])
        outs = ctx.msckf_update_sharded(capi.opts_from_scene(sc), comm, 0, 1)
        assert np.array_equal(outs["dx"], dx_plain) and np.array_equal(ctx.cov_download(), P_plain)
    finally:
        capi.rccl_comm_destroy(comm)
    # a pending pair dies with the covariance it was gated against: the next frame's update sees only its own batch
    ctx.cov_upload(sc.P)
    ctx.msckf_dense_blocks(sc.opts["chi2_mult"], blocks)
    ctx.cov_upload(sc.P)
    ctx.batch_upload_scene(sc, np.where(~dense)[0])
    out2 = ctx.msckf_update(capi.opts_from_scene(sc))
    ref2 = oracle.msckf_point_update(sc, feats=np.where(~dense)[0])
    assert np.abs(out2["dx"] - ref2["dx"]).max() < TOL_DX and relP(ctx.cov_download(), ref2["P"]) < TOL_P
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(C=8, F=60, seed=3, chi2_mult=1.0), dict(C=11, F=80, seed=5, chi2_mult=0.75, stereo_frac=0.3),
                                dict(C=20, F=40, seed=6, chi2_mult=1.0, stereo_frac=0.5)])
def test_updater_msckf_with_two_cameras(hiplib, oracle, kw):
    """UpdaterMSCKF::update on a state with TWO cameras (StateOptions::num_cameras = 2: extrinsics and intrinsics per camera,
    state/State.cpp:52-72): part of the features is seen by camera 0 only - they take the device batch (K1) - the others by both
    cameras at every clone (update/UpdaterHelper.cpp:335-344), which the batch format cannot carry: their rows are built by the host's
    get_feature_jacobian_full over both cameras, projected, and handed over as dense blocks (ovp_msckf_dense_blocks) that are gated
    against the resident covariance and join the SAME EKF update.  The third scene's stereo features have 40 measurements each -
    more than OVP_MAX_MEAS.  Reference: the dense numpy restatement of the whole update (np_ref.msckf_point_update_dense, pinned
    against the C oracle on one camera and by finite differences on two)."""
    from ov_plane_amd.build import build_host
    from ov_plane_amd.synth import make_stereo_scene, quat_boxplus
    from oracle import np_ref as R

    build_host()
    from ov_plane_amd import hostlib

    sc = make_stereo_scene(**kw)
    assert sc.n_stereo > 0 and sc.n_stereo < sc.F and int(sc.n_meas.max()) == 2 * sc.C
    tab = np.load(os.path.join(GOLD, "chi2_095_table.npy"))
    ref = R.msckf_point_update_dense(sc, tab)
    assert ref["accepted"].sum() >= 0.6 * sc.F and (kw["chi2_mult"] >= 1.0 or not ref["accepted"].all())
    out = hostlib.run_msckf_update(sc)
    assert (out["kept"] == ref["accepted"]).all() and out["deleted"].all() and not out["used"].any()
    dx = ref["dx"]
    ids = sc.ids
    for i in range(sc.C):
        cid = ids["clones"][i]
        assert np.abs(out["clone_q"][i] - quat_boxplus(sc.clone_q[i], dx[cid:cid + 3])).max() < TOL_DX
        assert np.abs(out["clone_p"][i] - (sc.clone_p[i] + dx[cid + 3:cid + 6])).max() < TOL_DX
    assert np.abs(out["calib_p"] - (sc.calib_p + dx[ids["calib"] + 3:ids["calib"] + 6])).max() < TOL_DX
    assert np.abs(out["intr"] - (sc.intr + dx[ids["intr"]:ids["intr"] + 8])).max() < TOL_DX
    c1 = out["cam1"]
    assert np.abs(c1["calib_q"] - quat_boxplus(sc.cam1["calib_q"], dx[ids["calib1"]:ids["calib1"] + 3])).max() < TOL_DX
    assert np.abs(c1["calib_p"] - (sc.cam1["calib_p"] + dx[ids["calib1"] + 3:ids["calib1"] + 6])).max() < TOL_DX
    assert np.abs(c1["intr"] - (sc.cam1["intr"] + dx[ids["intr1"]:ids["intr1"] + 8])).max() < TOL_DX
    assert np.abs(dx[ids["calib1"]:ids["calib1"] + 14]).max() > 1e-6     # camera 1's calibration really was corrected
    assert relP(out["P"], ref["P"]) < TOL_P


@pytest.mark.gpu
def test_updater_msckf_with_two_cameras_triangulates_first(hiplib, oracle):
    """The same with features that arrive WITHOUT positions (uvs_norm only, update/UpdaterMSCKF.cpp:120-166): the device
    triangulation knows camera 0, so a two-camera feature (and a track of more than 32 observations: the stereo features here have
    40) is triangulated from its first <= 32 observations of camera 0, then linearised over ALL of its measurements.  Reference:
    the oracle's triangulation of the camera-0 sub-tracks, then the dense numpy update at those positions."""
    from ov_plane_amd.build import build_host
    from ov_plane_amd.synth import Scene, make_stereo_scene, quat_boxplus
    from oracle import np_ref as R

    build_host()
    from ov_plane_amd import hostlib

    sc = make_stereo_scene(C=20, F=40, seed=6, chi2_mult=1.0, stereo_frac=0.5)
    C = sc.C
    mono = Scene(sc)     # camera 0's measurements of every feature: the first n_meas / 2 (stereo) or all of them
    m0 = np.where(np.arange(sc.F) < sc.n_stereo, sc.n_meas // 2, sc.n_meas).astype(np.int32)
    mono.update(n_meas=m0, uv=sc.uv[:, :C].copy(), uv_norm=sc.uv_norm[:, :C].copy(), clone_idx=sc.clone_idx[:, :C].copy())
    tri = oracle.triangulate(mono)
    assert tri["ok"].all()
    sc2 = Scene(sc)
    sc2["p_FinG"] = tri["p_FinG"]
    ref = R.msckf_point_update_dense(sc2, np.load(os.path.join(GOLD, "chi2_095_table.npy")))
    out = hostlib.run_msckf_update(sc, triangulate=True)
    assert (out["kept"] == ref["accepted"]).all() and ref["accepted"].sum() >= 30
    dx, ids = ref["dx"], sc.ids
    for i in range(C):
        cid = ids["clones"][i]
        assert np.abs(out["clone_q"][i] - quat_boxplus(sc.clone_q[i], dx[cid:cid + 3])).max() < TOL_DX
        assert np.abs(out["clone_p"][i] - (sc.clone_p[i] + dx[cid + 3:cid + 6])).max() < TOL_DX
    assert np.abs(out["cam1"]["intr"] - (sc.cam1["intr"] + dx[ids["intr1"]:ids["intr1"] + 8])).max() < TOL_DX
    assert relP(out["P"], ref["P"]) < TOL_P


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [
    dict(C=11, F=120, seed=71, chi2_mult=1.0),
    dict(C=10, F=150, seed=72, n_planes=4, feats_per_plane=20, chi2_mult=99999.0),
])
def test_updater_surface_takes_the_sharded_point_loop(hiplib, oracle, kw):
    """The plugin surface on several GPUs (call site core/VioManager.cpp:670): a State constructed on a given device
    (StateOptions::gpu_device) and UpdaterMSCKF::set_communicator -> update() runs the plane loop, then the point loop through
    ovp_msckf_update_sharded + ovp_rccl_gather_decisions.  With a communicator of one rank (the library's own RCCL binding) every
    output - state, covariance, feature-vector side effects - is bit-equal to the plain update() and equals the oracle."""
    from ov_plane_amd.build import build_host

    build_host()
    from ov_plane_amd import hostlib

    sc = make_scene(**kw)
    ref = _oracle_full_update(oracle, sc)
    plain = hostlib.run_msckf_update(sc)
    comm = hiplib.rccl_comm_create(hiplib.rccl_unique_id(), 0, 1, 0)
    try:
        out = hostlib.run_msckf_update(sc, comm=comm, rank=0, world=1, device=0)
    finally:
        hiplib.rccl_comm_destroy(comm)
    n_rest = int((~ref["used"]).sum())
    assert out["shard"] == (0, n_rest)          # the point batch of update() holds the leftovers only; one rank owns all of it
    for k in ("used", "kept", "deleted"):
        assert (out[k] == plain[k]).all()
    for k in ("clone_q", "clone_p", "calib_q", "calib_p", "intr", "cp_state", "P"):
        assert np.array_equal(out[k], plain[k]), k
    assert (out["used"] == ref["used"]).all() and (out["kept"] == ref["kept"]).all()
    assert np.abs(out["clone_p"] - ref["clone_p"]).max() < TOL_DX and relP(out["P"], ref["P"]) < TOL_P


def test_gather_decisions_is_a_no_op_on_one_rank_and_shares_tile_the_batch(hiplib):
    """ovp_rccl_gather_decisions on a one-rank communicator returns the arrays it was given; ovp_shard_range (context form) and
    ovp_shard_range_of_mask (pure form) agree behind a plane loop."""
    capi = hiplib
    sc = make_scene(C=12, F=260, seed=72, n_planes=5, feats_per_plane=30, planes_in_state_frac=0.6, chi2_mult=1.0)
    ctx = capi.Context(sc.N, sc.C, sc.F)
    ctx.cov_upload(sc.P)
    ctx.state_upload(sc)
    ctx.batch_upload_scene(sc)
    o = capi.opts_from_scene(sc)
    pl = ctx.plane_update(o, sc.plane_id, sc.cp, sc.cp_fej, sc.plane_state_id)
    o.skip_plane_used = 1
    for world in (1, 2, 3, 8):
        for rank in range(world):
            assert ctx.shard_range(o, rank, world) == capi.shard_range_of_mask(pl["used"], sc.F, rank, world)
    comm = capi.rccl_comm_create(capi.rccl_unique_id(), 0, 1, 0)
    try:
        out = ctx.msckf_update_sharded(o, comm, 0, 1)
        acc, chi2 = ctx.rccl_gather_decisions(comm, out["accepted"], out["chi2"])
        assert (acc == out["accepted"]).all() and np.array_equal(chi2, out["chi2"]) and acc.sum() > 10
    finally:
        capi.rccl_comm_destroy(comm)
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [
    dict(C=11, F=8, seed=5, ragged=True),
    dict(C=8, F=6, seed=6, ragged=True, chi2_mult=0.6),   # two candidates fail the gate: inert blocks, removed behind the loop
    dict(C=14, F=12, seed=7, ragged=True, do_fej=False),
    dict(C=9, F=5, seed=8, ragged=False, fisheye=True),
])
def test_slam_delayed_init_loop_on_the_device_matches_oracle(hiplib, oracle, kw):
    """ovp_slam_delayed_init (csrc/k_dinit.hip): the candidate loop of UpdaterSLAM::delayed_init (update/UpdaterSLAM.cpp:204-364) as
    one enqueue - rows at the device tables the previous candidate left, Householder split, chi2, initialize_invertible, update in
    place, commit - against ovo_slam_delayed_init: decisions, ids, landmark values, pose tables and covariance."""
    from ov_plane_amd.synth import quat_boxplus

    capi = hiplib
    sc = make_scene(**kw)
    ref = oracle.slam_delayed_init(sc)
    assert ref["ok"].any()
    if "chi2_mult" in kw:
        assert not ref["ok"].all()
    cap = sc.N + 3 * sc.F
    ctx = capi.Context(cap, sc.C, sc.F, device=0)
    ctx.cov_upload(sc.P)
    ctx.state_upload(sc)
    out = ctx.slam_delayed_init(capi.opts_from_scene(sc), sc.uv, sc.clone_idx, sc.n_meas, sc.p_FinG)
    assert (out["ok"] == ref["ok"]).all() and (out["new_id"] == ref["new_id"]).all()
    assert np.abs(out["chi2"] - ref["chi2"]).max() <= 1e-7 * max(1.0, np.abs(ref["chi2"]).max())
    assert ctx.cov_size() == ref["n"]
    # the caller's side: Type::update in order (landmark value = triangulated point + H_L^-1 res_init + every later correction)
    cq, cpos, intr = sc.clone_q.copy(), sc.clone_p.copy(), sc.intr.copy()
    p = sc.p_FinG.copy()
    for l in range(sc.F):
        if not out["ok"][l]:
            continue
        dx = out["dx"][l]
        p[l] += out["delta_init"][l]
        for g in range(l + 1):
            if out["ok"][g]:
                i = int(out["new_id"][g])
                p[g] += dx[i:i + 3]
        for i in range(sc.C):
            cid = sc.ids["clones"][i]
            cq[i] = quat_boxplus(cq[i], dx[cid:cid + 3])
            cpos[i] = cpos[i] + dx[cid + 3:cid + 6]
        intr = intr + dx[22:30]
    ok = ref["ok"]
    assert np.abs(p[ok] - ref["p"][ok]).max() < TOL_DX
    assert np.abs(cpos - ref["clone_p"]).max() < TOL_DX and np.abs(cq - ref["clone_q"]).max() < TOL_DX
    assert np.abs(intr - ref["intr"]).max() < TOL_DX
    assert relP(ctx.cov_download(), ref["P"]) < TOL_P
    ctx.close()


@pytest.mark.gpu
def test_round5_entry_points_edge_cases(hiplib, oracle):
    """Empty and ragged inputs of the round-5 entries: no landmarks, a landmark without observations among others (left alone like
    the reference's clean-up would, update/UpdaterSLAM.cpp:412-415), a delayed initialisation that would outgrow the context
    (OVP_E_CAPACITY, nothing touched), and index ranges of a sharded update that tile the batch - an empty share included."""
    from ov_plane_amd.synth import make_slam_scene

    capi = hiplib
    sc = make_slam_scene(C=8, n_slam=6, seed=21)
    ctx = capi.Context(sc.N, sc.C, sc.F, device=0)
    ctx.cov_upload(sc.P)
    ctx.state_upload(sc)
    o = capi.opts_from_scene(sc)
    # no landmarks at all
    out = ctx.slam_update(o, np.zeros((0, 1, 2), np.float32), np.zeros((0, 1), np.int32), np.zeros(0, np.int32), np.zeros((0, 3)),
                          np.zeros((0, 3)), np.zeros(0, np.int32))
    assert out["rc"] == 0 and np.abs(out["dx"]).max() == 0.0 and np.array_equal(ctx.cov_download(), sc.P)
    # one landmark lost all its observations: it takes no part, the others are updated as if it were not in the call
    nm = sc.n_meas.copy()
    nm[2] = 0
    keep = np.array([0, 1, 3, 4, 5])
    a = ctx.slam_update(o, sc.uv, sc.clone_idx, nm, sc.p_FinG, sc.p_FinG_fej, sc.lm_id)
    Pa = ctx.cov_download()
    ctx.cov_upload(sc.P)
    b = ctx.slam_update(o, sc.uv[keep], sc.clone_idx[keep], sc.n_meas[keep], sc.p_FinG[keep], sc.p_FinG_fej[keep], sc.lm_id[keep])
    assert a["status"][2] == 0 and (a["status"][keep] == b["status"]).all() and b["status"].all()
    assert np.abs(a["dx"] - b["dx"]).max() < 1e-12 and relP(Pa, ctx.cov_download()) < 1e-12
    ctx.close()
    # delayed initialisation: no room for the new landmarks -> OVP_E_CAPACITY and an untouched covariance
    sc2 = make_scene(C=8, F=5, seed=6, ragged=True)
    ctx = capi.Context(sc2.N + 6, sc2.C, sc2.F, device=0)   # room for two of the five candidates
    ctx.cov_upload(sc2.P)
    ctx.state_upload(sc2)
    r = ctx.slam_delayed_init(capi.opts_from_scene(sc2), sc2.uv, sc2.clone_idx, sc2.n_meas, sc2.p_FinG, raise_on_error=False)
    assert r["rc"] == capi.OVP_E_CAPACITY and ctx.cov_size() == sc2.N and np.array_equal(ctx.cov_download(), sc2.P)
    ok2 = ctx.slam_delayed_init(capi.opts_from_scene(sc2), sc2.uv[:2], sc2.clone_idx[:2], sc2.n_meas[:2], sc2.p_FinG[:2])
    assert ok2["rc"] == 0 and ctx.cov_size() == sc2.N + 3 * int(ok2["ok"].sum())
    ctx.close()
    # sharded update without a communicator: the shares of the ranks tile the batch, a rank may get nothing
    sc3 = make_scene(C=6, F=5, seed=9, chi2_mult=1.0)
    ctx = capi.Context(sc3.N, sc3.C, sc3.F, device=0)
    ctx.state_upload(sc3)
    ctx.batch_upload_scene(sc3)
    seen = np.zeros(sc3.F, dtype=int)
    o3 = capi.opts_from_scene(sc3)
    for rank in range(7):
        lo, hi = ctx.shard_range(o3, rank, 7)
        seen[lo:hi] += 1
        ctx.cov_upload(sc3.P)
        ctx.batch_set_range(lo, hi)
        out = ctx.msckf_update(o3)
        assert not out["accepted"][:lo].any() and not out["accepted"][hi:].any()
        if hi == lo:
            assert np.abs(out["dx"]).max() == 0.0 and relP(ctx.cov_download(), sc3.P) < 1e-12
    ctx.batch_set_range(-1, -1)
    ctx.cov_upload(sc3.P)
    one = ctx.msckf_update_sharded(o3, None, 0, 1)
    assert one["shard"] == (0, sc3.F) and one["accepted"].sum() >= 3
    assert (seen == 1).all()
    ctx.close()
