#!/usr/bin/env python
"""CPU study of the reference's plane-level chi2 (update/UpdaterMSCKF.cpp:588-631 on the system UpdaterPlane.cpp:519-552 leaves):
taps the oracle's stacked system before the Givens compression and the system it hands to the test, and splits the statistic
into the part inside range(H) and the energy of the retained rows that carry no Jacobian content.  Test infrastructure."""
from __future__ import annotations

import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

TAP = C.CFUNCTYPE(None, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double),
                  C.POINTER(C.c_double))


def tap_planes(sc, libpath=None):
    from oracle import pyoracle
    L = pyoracle.lib() if libpath is None else C.CDLL(libpath)
    got = {}

    def cb(pl, stage, rows, cols, ld, Hx, Hcp, res):
        H = np.ctypeslib.as_array(Hx, shape=(cols, ld))[:, :rows].T.copy()
        r = np.ctypeslib.as_array(res, shape=(rows,)).copy()
        if stage == 0:
            Hc = np.ctypeslib.as_array(Hcp, shape=(3, ld))[:, :rows].T.copy()
            got.setdefault(pl, {})["pre"] = (H, Hc, r)
        else:
            Pm = np.ctypeslib.as_array(Hcp, shape=(cols, cols)).copy()
            got.setdefault(pl, {})["post"] = (H, Pm, r)

    fn = TAP(cb)
    L.ovo_set_plane_tap(fn)
    try:
        ref = pyoracle.msckf_plane_update(sc, libpath=libpath)
    finally:
        L.ovo_set_plane_tap(TAP())
    return ref, got


def split(H, Pm, r, tol=1e-9):
    """chi2 of (H, r) under Pm and its parts: inside range(H) / outside."""
    U, s, Vt = np.linalg.svd(H, full_matrices=False)
    k = int((s > tol * s[0]).sum())
    U1 = U[:, :k]
    y = U1.T @ r
    S1 = (np.diag(s[:k]) @ Vt[:k]) @ Pm @ (np.diag(s[:k]) @ Vt[:k]).T + np.eye(k)
    t1 = float(y @ np.linalg.solve(S1, y))
    t2 = float(r @ r - y @ y)
    S = H @ Pm @ H.T + np.eye(H.shape[0])
    chi2 = float(r @ np.linalg.solve(S, r))
    return dict(rank=k, t1=t1, t2=t2, chi2=chi2, sv=s)


if __name__ == "__main__":
    from ov_plane_amd.synth import make_scene
    seeds = [int(a) for a in sys.argv[1:]] or [100]
    for seed in seeds:
        sc = make_scene(C=30, F=2000, seed=seed, n_planes=20, feats_per_plane=50, planes_in_state_frac=0.5, chi2_mult=1.0)
        ref, got = tap_planes(sc)
        for pl in sorted(got):
            H0, Hcp0, r0 = got[pl]["pre"]
            H1, Pm, r1 = got[pl]["post"]
            sp = split(H1, Pm, r1)
            M, c = H0.shape
            in_state = sc.plane_state_id[pl] >= 0
            # exact quantities on the uncompressed system
            Hall = np.hstack([H0, Hcp0])
            sv_all = np.linalg.svd(Hall, compute_uv=False)
            sv_x = np.linalg.svd(H0, compute_uv=False)
            print(json.dumps(dict(seed=seed, pl=pl, in_state=bool(in_state), M=M, c=c, rows_u=H1.shape[0], chi2_ref=float(ref["plane_chi2"][pl]),
                                  chi2_np=sp["chi2"], rank=sp["rank"], t1=sp["t1"], t2=sp["t2"], rr=float(r0 @ r0),
                                  sv_tail=[float("%.3g" % v) for v in sp["sv"][-16:]],
                                  svx_tail=[float("%.3g" % v) for v in sv_x[-14:]],
                                  svall_tail=[float("%.3g" % v) for v in sv_all[-14:]])))
