"""Runs the plane loop many times on the same frames and compares dx, the decisions and the covariance bitwise: the hand-over
protocols inside k_chol2 (LDS counters between elimination, tile and chain waves) must not leave a timing-dependent result."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ov_plane_amd import capi
from ov_plane_amd.synth import make_scene
bad = 0
for seed, kw in ((0, dict(C=30, F=2000, n_planes=20, feats_per_plane=50)), (3, dict(C=30, F=4000, n_planes=30, feats_per_plane=60)), (5, dict(C=11, F=600, n_planes=8, feats_per_plane=30))):
    sc = make_scene(seed=seed, planes_in_state_frac=0.5, chi2_mult=1.0, **kw)
    ctx = capi.Context(sc.N, sc.C, sc.F, device=0)
    o = capi.opts_from_scene(sc)
    ref = None
    for rep in range(150):
        ctx.cov_upload(sc.P); ctx.state_upload(sc); ctx.batch_upload_scene(sc)
        pl = ctx.plane_update(o, sc.plane_id, sc.cp, sc.cp_fej, sc.plane_state_id)
        P = ctx.cov_download()
        key = (pl["dx"].tobytes(), pl["ok"].tobytes(), P.tobytes())
        if ref is None: ref = key
        elif key != ref:
            bad += 1
    print("seed", seed, "N", sc.N, "accepted", int(pl["ok"].sum()), "mismatches so far", bad)
    ctx.close()
print("DETERMINISM", "OK" if bad == 0 else "FAILED %d" % bad)
