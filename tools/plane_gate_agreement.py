#!/usr/bin/env python
"""Plane-level chi2 gate (update/UpdaterMSCKF.cpp:607-631) at chi2_multipler = 1: device statistic against the oracle's, next to
the distance between two builds of the oracle itself.

For every seed: the oracle runs the plane loop of a config-3 sized scene (30 clones, 20 planes x 50 features, half the planes in
the state) with the real gate.  The device runs the same loop with the oracle's accept / reject sequence forced
(ovp_plane_batch::force_decision), and so does a second build of the oracle compiled with fused multiply-adds
(oracle/Makefile: fma), so all three see the same state and covariance at every plane and the statistics are compared plane by
plane: value difference (overall, in-state / out-of-state planes), the decision each gate would have taken, distance of the
disagreements from the threshold.  `interbuild_band` = the largest |chi2_fma - chi2_plain| observed: how far the reference's own
statistic moves when nothing but the compiler's contraction of a*b+c changes.  Prints one JSON object (committed under profiles/)."""
from __future__ import annotations

import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402


def run(seeds, C=30, F=2000, n_planes=20, feats_per_plane=50, chi2_mult=1.0, verbose=False, fma=True, device=True):
    from oracle import pyoracle
    from ov_plane_amd.synth import make_scene

    fma_so = pyoracle.build_fma() if fma else None
    if device:
        from ov_plane_amd import capi
    rows = []
    ctx = None
    for seed in seeds:
        try:
            sc = make_scene(C=C, F=F, seed=seed, n_planes=n_planes, feats_per_plane=feats_per_plane, planes_in_state_frac=0.5,
                            chi2_mult=chi2_mult)
        except RuntimeError:  # the generator could not place every feature in view for this seed
            continue
        ref = pyoracle.msckf_plane_update(sc)
        alt = pyoracle.msckf_plane_update(sc, libpath=fma_so, force=ref["plane_ok"]) if fma_so else None
        relP = 0.0
        out = None
        if device:
            if ctx is None:
                ctx = capi.Context(sc.N, sc.C, sc.F)
            ctx.cov_upload(sc.P)
            ctx.state_upload(sc)
            ctx.batch_upload_scene(sc)
            o = capi.opts_from_scene(sc)
            out = ctx.plane_update(o, sc.plane_id, sc.cp, sc.cp_fej, sc.plane_state_id, force_decision=ref["plane_ok"].astype(np.uint8))
            P = ctx.cov_download()
            d = np.sqrt(np.abs(np.diag(ref["P"])))
            relP = float((np.abs(P - ref["P"]) / np.outer(d, d)).max())
        for k in range(n_planes):
            if ref["plane_rows"][k] <= 0:
                continue
            dof = int(ref["plane_rows"][k])
            thr = chi2_mult * pyoracle.lib().ovo_chi2_quantile_095(dof)
            r = dict(seed=int(seed), plane=k, in_state=bool(sc.plane_state_id[k] >= 0), dof_ref=dof, thr=float(thr),
                     chi2_ref=float(ref["plane_chi2"][k]), ok_ref=bool(ref["plane_ok"][k]), relP=relP)
            if out is not None:
                r.update(dof=int(out["dof"][k]), chi2_dev=float(out["chi2"][k]), ok_dev=bool(out["chi2"][k] <= thr))
            if alt is not None:
                r.update(chi2_fma=float(alt["plane_chi2"][k]), ok_fma=bool(alt["plane_chi2"][k] <= thr))
            rows.append(r)
        if verbose:
            print("seed %d: oracle accepted %d/%d, relP %.2e" % (seed, int(ref["plane_ok"].sum()), n_planes, relP), file=sys.stderr)
    if ctx is not None:
        ctx.close()
    return rows


def _stats(d):
    d = np.asarray(d, dtype=float)
    if d.size == 0:
        return dict(n=0)
    return dict(n=int(d.size), mean=float(d.mean()), sem=float(d.std() / np.sqrt(d.size)), std=float(d.std()),
                abs_max=float(np.abs(d).max()))


def summarise(rows):
    out = dict(planes=len(rows), seeds=len({r["seed"] for r in rows}), oracle_accept_rate=float(np.mean([r["ok_ref"] for r in rows])))
    if rows and "chi2_fma" in rows[0]:
        db = [r["chi2_fma"] - r["chi2_ref"] for r in rows]
        flips = [r for r in rows if r["ok_fma"] != r["ok_ref"]]
        out.update(interbuild=_stats(db), interbuild_band=float(np.abs(db).max()), interbuild_flips=len(flips),
                   interbuild_flip_margin_max=float(max([abs(r["chi2_ref"] - r["thr"]) for r in flips], default=0.0)))
    if rows and "chi2_dev" in rows[0]:
        d = np.array([r["chi2_dev"] - r["chi2_ref"] for r in rows])
        dis = [r for r in rows if r["ok_ref"] != r["ok_dev"]]
        margin = [abs(r["chi2_ref"] - r["thr"]) for r in dis]
        out.update(disagreements=len(dis), disagreement_rate=len(dis) / max(len(rows), 1),
                   disagreements_oracle_rejects=int(sum(not r["ok_ref"] for r in dis)),
                   diff_mean=float(d.mean()), diff_std=float(d.std()), diff_abs_max=float(np.abs(d).max()),
                   diff_in_state=_stats([r["chi2_dev"] - r["chi2_ref"] for r in rows if r["in_state"]]),
                   diff_out_of_state=_stats([r["chi2_dev"] - r["chi2_ref"] for r in rows if not r["in_state"]]),
                   disagreement_margin_max=float(max(margin)) if margin else 0.0,
                   dof_mismatch=int(sum(r["dof"] != r["dof_ref"] for r in rows)),
                   relP_max=float(max(r["relP"] for r in rows)))
        if "interbuild_band" in out:
            out["disagreements_outside_interbuild_band"] = int(sum(m > out["interbuild_band"] for m in margin))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=50)
    ap.add_argument("--first-seed", type=int, default=100)
    ap.add_argument("--rows", action="store_true", help="include the per-plane table")
    ap.add_argument("--no-device", action="store_true", help="oracle builds only (runs without a GPU)")
    args = ap.parse_args()
    rows = run(range(args.first_seed, args.first_seed + args.seeds), verbose=True, device=not args.no_device)  # a few seeds are skipped
    out = summarise(rows)
    if args.rows:
        out["rows"] = rows
    print(json.dumps(out))


if __name__ == "__main__":
    main()
