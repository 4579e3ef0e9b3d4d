"""In-kernel phase stamps of K1 (library built with -DOVP_K1_STAMPS, passed as OVP_LIB_AB): cycles per phase of a feature wave,
median over the features of a full-length-track batch.  usage: OVP_LIB_AB=path/to/stamped.so python tools/k1_stamps.py [C F]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ov_plane_amd import capi  # noqa: E402
from ov_plane_amd.synth import make_scene  # noqa: E402

C_ = int(sys.argv[1]) if len(sys.argv) > 1 else 30
F_ = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
sc = make_scene(C=C_, F=F_, seed=0, chi2_mult=1.0)
for scratch in (False, True):
    if scratch:
        os.environ["OVP_K1_BSCR"] = "1"
    else:
        os.environ.pop("OVP_K1_BSCR", None)
    ctx = capi.Context(sc.N, sc.C, sc.F)
    ctx.state_upload(sc)
    ctx.batch_upload_scene(sc, None)
    ctx.debug_read("cycles_on", (1,))
    o = capi.opts_from_scene(sc)
    for it in range(3):
        ctx.cov_upload(sc.P)
        ctx.msckf_update(o)
    cyc = ctx.debug_read("cycles", (sc.F * 10,), dtype=np.int64)
    st = cyc[: sc.F * 8].reshape(sc.F, 8)
    ex = cyc[sc.F * 8: sc.F * 10].reshape(sc.F, 2)
    d = np.diff(st, axis=1)
    names = ["A rows", "A2 e,u", "B (scratch form)", "C factor" if scratch else "B+C blocks", "D/gate", "E projector", "stores"]
    print("== %s, C=%d F=%d: cycles per feature wave (median / max), total %.0f / %.0f" %
          ("B through scratch" if scratch else "B in LDS", C_, F_, np.median(st[:, 7] - st[:, 0]), (st[:, 7] - st[:, 0]).max()))
    for k, nm in enumerate(names):
        print("   %-18s %8.0f %8.0f" % (nm, np.median(d[:, k]), d[:, k].max()))
    if not scratch:
        print("   inside the blocks: build %.0f, elimination %.0f (median)" % (np.median(ex[:, 0]), np.median(ex[:, 1])))
    tot = (st[:, 7] - st[:, 0]).astype(float)
    print("   total, percentiles 5/25/50/75/95/100: %s" % " ".join("%.0f" % np.percentile(tot, q) for q in (5, 25, 50, 75, 95, 100)))
    nwg = sc.F if sc.F <= 255 else (255 if sc.F <= 8 * 255 else (sc.F + 7) // 8)
    f = np.arange(sc.F)
    wave, wg = f // nwg, f % nwg
    print("   mean total by wave slot of the workgroup: %s" % " ".join("%.0f" % tot[wave == w].mean() for w in range(wave.max() + 1)))
    print("   mean total by workgroup index mod 8 (XCD turn): %s" % " ".join("%.0f" % tot[wg % 8 == x].mean() for x in range(8)))
    print("   mean start offset by wave slot (cycles after the earliest start of the same counter domain is not comparable across XCDs)")
    for k, nm in enumerate(names):
        dd = d[:, k].astype(float)
        print("   %-18s p5 %7.0f p95 %7.0f  by wave slot: %s" % (nm, np.percentile(dd, 5), np.percentile(dd, 95),
              " ".join("%.0f" % dd[wave == w].mean() for w in range(wave.max() + 1))))
    ctx.close()
