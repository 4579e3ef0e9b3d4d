#!/usr/bin/env python
"""Plane-level chi2 gate (update/UpdaterMSCKF.cpp:607-631) at chi2_multipler = 1: the device statistic against an ENSEMBLE of the
oracle's - four roundings of the same restatement (tests/golden/plane_gate_ensemble.npz, made by
tests/golden/make_plane_gate_ensemble.py: plain, fma, x87, assoc; oracle/Makefile).

The reference's statistic contains rows of a rank-deficient Givens sweep whose content is decided by rounding (NOTES.md 3b), so
one oracle run is one sample of "what the reference answers"; the builds of the fixture all ran with the plain build's accept /
reject sequence imposed, and so does the device here (ovp_plane_batch::force_decision): all five see the same state and
covariance at every plane and the statistics compare plane by plane.

Reported (one JSON object, committed under profiles/):
  * oracle against oracle: largest distance between two builds on one plane (`interbuild_band`), flip rate of every pair of builds;
  * device against every build and against the ensemble mean: mean / spread of the difference, overall and for in-state /
    out-of-state planes (the bias the verdict of round 5 asked to remove is the mean against the ENSEMBLE MEAN, whose own noise
    is half a single build's);
  * decisions: flip rate of the device against every build, against the majority; planes where the builds are unanimous and the
    device is not with them (`unanimous_violations`, each with its margins) - the quantity tests/test_gpu_parity.py bounds.
Needs a GPU (device part); --no-device prints the oracle-only part."""
from __future__ import annotations

import argparse
import itertools
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

FIXTURE = os.path.join(ROOT, "tests", "golden", "plane_gate_ensemble.npz")


def load_fixture(path=FIXTURE):
    z = np.load(path)
    fx = {k: z[k] for k in z.files}
    fx["scenes"] = [json.loads(str(s)) for s in fx["scenes"]]
    fx["builds"] = [str(b) for b in fx["builds"]]
    return fx


def device_statistics(capi, fx, scene_indices=None, verbose=False):
    """chi2 / dof of the device's plane loop on every scene of the fixture, run on the plain build's decisions.  Returns arrays
    aligned with the fixture's per-plane rows (NaN / -1 for scenes not run) and the largest covariance-independent check: the
    decisions the device reports equal the imposed ones."""
    from ov_plane_amd.synth import make_scene

    n = len(fx["scene"])
    chi2 = np.full(n, np.nan)
    dof = np.full(n, -1, dtype=np.int64)
    ctxs = {}
    todo = range(len(fx["scenes"])) if scene_indices is None else scene_indices
    for s in todo:
        kw = fx["scenes"][s]
        sc = make_scene(**kw)
        rows = np.where(fx["scene"] == s)[0]
        assert (fx["plane"][rows] == np.arange(len(rows))).all()
        key = (sc.N, sc.C, sc.F)
        if key not in ctxs:
            ctxs[key] = capi.Context(sc.N, sc.C, sc.F)
        ctx = ctxs[key]
        ctx.cov_upload(sc.P)
        ctx.state_upload(sc)
        ctx.batch_upload_scene(sc)
        o = capi.opts_from_scene(sc)
        out = ctx.plane_update(o, sc.plane_id, sc.cp, sc.cp_fej, sc.plane_state_id, force_decision=fx["ok"][rows].astype(np.uint8))
        assert (np.asarray(out["ok"]).astype(bool) == fx["ok"][rows]).all()
        chi2[rows] = out["chi2"]
        dof[rows] = out["dof"]
        if verbose:
            print("scene %d (%s): %d planes" % (s, kw, len(rows)), file=sys.stderr)
    for c in ctxs.values():
        c.close()
    return chi2, dof


def _stats(d):
    d = np.asarray(d, dtype=float)
    if d.size == 0:
        return dict(n=0)
    return dict(n=int(d.size), mean=float(d.mean()), sem=float(d.std() / np.sqrt(d.size)), std=float(d.std()), abs_max=float(np.abs(d).max()))


def analyse(fx, chi2_dev=None, dof_dev=None):
    live = fx["dof"] > 0
    if chi2_dev is not None:
        live = live & np.isfinite(chi2_dev)
    E = fx["chi2"][live]
    thr = fx["thr"][live]
    ins = fx["in_state"][live]
    builds = fx["builds"]
    dec = E <= thr[:, None]
    una = dec.all(axis=1) | (~dec).all(axis=1)
    out = dict(planes=int(live.sum()), scenes=int(len(set(fx["scene"][live].tolist()))), builds=builds,
               oracle_accept_rate=float(dec[:, 0].mean()), ensemble_unanimous=int(una.sum()), ensemble_unanimous_rate=float(una.mean()),
               interbuild_band=float((E.max(axis=1) - E.min(axis=1)).max()))
    pairs = {}
    for a, b in itertools.combinations(range(len(builds)), 2):
        d = E[:, b] - E[:, a]
        pairs["%s-%s" % (builds[b], builds[a])] = dict(_stats(d), flips=int((dec[:, a] != dec[:, b]).sum()),
                                                      flip_rate=float((dec[:, a] != dec[:, b]).mean()))
    out["oracle_vs_oracle"] = pairs
    out["oracle_vs_oracle_flip_rate_mean"] = float(np.mean([p["flip_rate"] for p in pairs.values()]))
    if chi2_dev is None:
        return out
    D = chi2_dev[live]
    ddec = D <= thr
    out["dof_mismatch"] = int((dof_dev[live] != fx["dof"][live]).sum())
    out["device_vs_build"] = {}
    for a, nm in enumerate(builds):
        d = D - E[:, a]
        out["device_vs_build"][nm] = dict(_stats(d), flips=int((ddec != dec[:, a]).sum()), flip_rate=float((ddec != dec[:, a]).mean()))
    out["device_vs_build_flip_rate_mean"] = float(np.mean([v["flip_rate"] for v in out["device_vs_build"].values()]))
    dm = D - E.mean(axis=1)
    out["device_vs_ensemble_mean"] = dict(all=_stats(dm), in_state=_stats(dm[ins]), out_of_state=_stats(dm[~ins]))
    maj = dec.sum(axis=1) * 2 >= dec.shape[1]
    out["device_vs_majority_flips"] = int((ddec != maj).sum())
    out["device_vs_majority_flip_rate"] = float((ddec != maj).mean())
    # the contract: wherever the builds are unanimous the device is with them
    bad = np.where(una & (ddec != dec[:, 0]))[0]
    idx = np.where(live)[0]
    out["unanimous_violations"] = [dict(scene=int(fx["scene"][idx[k]]), plane=int(fx["plane"][idx[k]]), thr=float(thr[k]), chi2_device=float(D[k]),
                                        chi2_builds=[float(v) for v in E[k]], device_margin=float(abs(D[k] - thr[k])),
                                        nearest_build_margin=float(np.abs(E[k] - thr[k]).min())) for k in bad]
    out["unanimous_violation_rate"] = float(len(bad) / max(int(una.sum()), 1))
    out["device_inside_build_range"] = float(((D >= E.min(axis=1)) & (D <= E.max(axis=1))).mean())
    out["device_to_nearest_build_abs_max"] = float(np.minimum(np.abs(D[:, None] - E).min(axis=1), 1e300).max())
    out["device_to_plain_abs_max"] = float(np.abs(D - E[:, 0]).max())
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--no-device", action="store_true", help="oracle builds only (runs without a GPU)")
    ap.add_argument("--rows", action="store_true", help="include the per-plane table")
    ap.add_argument("--note", default="")
    ap.add_argument("--fit", action="store_true",
                    help="study: the device loop twice, with the expected energy of the rounding-decided rows weighted 0 and 1 "
                         "(OVP_PL_NOISE_SCALE), and the weight that removes the bias against the ensemble mean - fitted on the 50 "
                         "scenes of the first family, checked on the ten others")
    args = ap.parse_args()
    fx = load_fixture()
    chi2_dev = dof_dev = None
    if args.fit:
        from ov_plane_amd import capi

        os.environ["OVP_PL_NOISE_SCALE"] = "0"
        c0, _ = device_statistics(capi, fx)
        os.environ["OVP_PL_NOISE_SCALE"] = "1"
        c1, _ = device_statistics(capi, fx)
        del os.environ["OVP_PL_NOISE_SCALE"]
        t2 = c1 - c0                                   # expected energy of the rounding-decided rows, per plane
        Em = fx["chi2"].mean(axis=1)
        fam1 = fx["scene"] < 50
        fit = {}
        for nm, sel in (("all_first_family", fam1), ("in_state", fam1 & fx["in_state"]), ("out_of_state", fam1 & ~fx["in_state"])):
            k = float(((Em - c0)[sel] * t2[sel]).sum() / (t2[sel] ** 2).sum())         # least squares through the origin
            k_mean = float((Em - c0)[sel].mean() / t2[sel].mean())                      # the weight that zeroes the mean difference
            fit[nm] = dict(kappa_least_squares=k, kappa_zero_mean=k_mean, term_mean=float(t2[sel].mean()), n=int(sel.sum()))
        k = fit["all_first_family"]["kappa_zero_mean"]
        val = {}
        for nm, sel in (("first_family", fam1), ("held_out_config3_config4", ~fam1)):
            val[nm] = dict(before=_stats((c1 - Em)[sel]), after=_stats((c0 + k * t2 - Em)[sel]))
        print(json.dumps(dict(fit=fit, kappa=k, bias_against_ensemble_mean=val)))
        return
    if not args.no_device:
        from ov_plane_amd import capi

        chi2_dev, dof_dev = device_statistics(capi, fx, verbose=True)
    out = analyse(fx, chi2_dev, dof_dev)
    if args.note:
        out["note"] = args.note
    if args.rows and chi2_dev is not None:
        out["rows"] = [dict(scene=int(s), plane=int(p), in_state=bool(i), dof=int(d), thr=float(t), chi2_device=float(c), chi2_builds=[float(v) for v in e])
                       for s, p, i, d, t, c, e in zip(fx["scene"], fx["plane"], fx["in_state"], fx["dof"], fx["thr"], chi2_dev, fx["chi2"])]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
