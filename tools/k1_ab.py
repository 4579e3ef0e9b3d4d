"""K1 A/B on the GPU box: the bordered factorization with B built inside the factorization (default, round 5) against the form that
sends B through the per-feature scratch in device memory (OVP_K1_BSCR=1), and both against the oracle, feature by feature.
usage: python tools/k1_ab.py [--time]    (prints a table; exit code 1 on a mismatch)"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ov_plane_amd.synth import make_scene  # noqa: E402


def run(capi, sc, scratch):
    if scratch:
        os.environ["OVP_K1_BSCR"] = "1"
    else:
        os.environ.pop("OVP_K1_BSCR", None)
    ctx = capi.Context(sc.N, sc.C, max(sc.F, 1))
    ctx.cov_upload(sc.P)
    ctx.state_upload(sc)
    ctx.batch_upload_scene(sc, None)
    out = ctx.msckf_update(capi.opts_from_scene(sc))
    out["P"] = ctx.cov_download()
    ctx.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--time", action="store_true")
    ap.add_argument("--time-feats", type=int, default=2000)
    args = ap.parse_args()
    from ov_plane_amd.build import build_lib

    build_lib()
    from ov_plane_amd import capi
    from oracle import pyoracle

    pyoracle.build()
    bad = 0
    cases = [
        dict(C=5, F=40, seed=1, ragged=True, min_meas=2, chi2_mult=1.0),
        dict(C=9, F=120, seed=2, ragged=True, min_meas=2, chi2_mult=1.0),
        dict(C=16, F=200, seed=3, ragged=True, min_meas=2, chi2_mult=1.0),
        dict(C=23, F=300, seed=4, ragged=True, min_meas=2, chi2_mult=1.0),
        dict(C=30, F=400, seed=5, ragged=True, min_meas=2, chi2_mult=1.0),
        dict(C=30, F=100, seed=6, chi2_mult=1.0),
        dict(C=30, F=64, seed=7, chi2_mult=0.7),
    ]
    for kw in cases:
        sc = make_scene(**kw)
        ref = pyoracle.msckf_point_update(sc)
        new = run(capi, sc, False)
        old = run(capi, sc, True)
        m = np.asarray(sc.n_meas)
        scale = np.maximum(1.0, np.abs(ref["chi2"]))
        e_new = np.abs(new["chi2"] - ref["chi2"]) / scale
        e_old = np.abs(old["chi2"] - ref["chi2"]) / scale
        acc_new = int((new["accepted"] != ref["accepted"]).sum())
        acc_old = int((old["accepted"] != ref["accepted"]).sum())
        dxe = float(np.abs(new["dx"] - ref["dx"]).max())
        print("C=%d F=%d seed=%d: chi2 rel err new %.2e old %.2e | accept mismatches new %d old %d | dx err new %.2e" %
              (kw["C"], kw["F"], kw["seed"], np.nanmax(e_new), np.nanmax(e_old), acc_new, acc_old, dxe))
        if not (np.nanmax(e_new) <= 1e-8) or acc_new or np.isnan(new["chi2"]).sum() != np.isnan(ref["chi2"]).sum():
            bad += 1
            for mm in sorted(set(m.tolist())):
                sel = m == mm
                print("   m=%2d: %3d features, worst chi2 rel err new %.3e old %.3e, first chi2 new/ref %s / %s" %
                      (mm, sel.sum(), np.nanmax(e_new[sel]), np.nanmax(e_old[sel]), new["chi2"][sel][:2], ref["chi2"][sel][:2]))
    if args.time:
        sc = make_scene(C=30, F=args.time_feats, seed=0, chi2_mult=1.0)
        for scratch in (True, False, True, False):
            if scratch:
                os.environ["OVP_K1_BSCR"] = "1"
            else:
                os.environ.pop("OVP_K1_BSCR", None)
            ctx = capi.Context(sc.N, sc.C, sc.F)
            ctx.state_upload(sc)
            ctx.batch_upload_scene(sc, None)
            o = capi.opts_from_scene(sc)
            ts = []
            for it in range(30):
                ctx.cov_upload(sc.P)
                t0 = time.perf_counter()
                ctx.msckf_update(o)
                ts.append(time.perf_counter() - t0)
            ctx.close()
            print("config-2 update (%d features), %s: median %.1f us (min %.1f)" % (sc.F, "B through scratch" if scratch else "B in LDS", 1e6 * np.median(ts[5:]), 1e6 * min(ts[5:])))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
