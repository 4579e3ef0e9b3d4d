#!/usr/bin/env python
"""Runs the BASELINE config-3 / config-4 step (plane loop + point update) a number of times: the command rocprofv3 wraps for the
plane-loop kernel statistics under profiles/.  Prints the host-side time of the two halves."""
from __future__ import annotations

import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clones", type=int, default=30)
    ap.add_argument("--feats", type=int, default=2000)
    ap.add_argument("--planes", type=int, default=20)
    ap.add_argument("--feats-per-plane", type=int, default=50)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--chi2-mult", type=float, default=1.0)
    args = ap.parse_args()
    from ov_plane_amd import capi
    from ov_plane_amd.synth import make_scene

    sc = make_scene(C=args.clones, F=args.feats, seed=0, n_planes=args.planes, feats_per_plane=args.feats_per_plane,
                    planes_in_state_frac=0.5, chi2_mult=args.chi2_mult)
    ctx = capi.Context(sc.N, sc.C, sc.F, device=0)
    o = capi.opts_from_scene(sc)
    tt = []
    for it in range(args.reps + 3):
        ctx.cov_upload(sc.P)
        ctx.state_upload(sc)
        ctx.batch_upload_scene(sc)
        t0 = time.perf_counter()
        pl = ctx.plane_update(o, sc.plane_id, sc.cp, sc.cp_fej, sc.plane_state_id)
        t1 = time.perf_counter()
        ctx.batch_upload_scene(sc, np.where(~pl["used"])[0])
        t2 = time.perf_counter()
        pt = ctx.msckf_update(o)
        t3 = time.perf_counter()
        if it >= 3:
            tt.append((t1 - t0, t3 - t2))
    tt = 1e3 * np.array(tt).mean(axis=0)
    print("N=%d planes accepted %d/%d, points accepted %d; plane loop %.3f ms, point update %.3f ms" % (
        sc.N, int(pl["ok"].sum()), args.planes, int(pt["accepted"].sum()), tt[0], tt[1]))
    ctx.close()


if __name__ == "__main__":
    main()
