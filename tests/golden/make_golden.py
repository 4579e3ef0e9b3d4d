"""Generates the committed fixtures under tests/golden/ (run from the repo root: python tests/golden/make_golden.py).

 * chi2_095_table.npy : scipy.stats.chi2.ppf(0.95, k), k = 0..1000 (k=0 -> 0), the pin for the boost quantile the
                        reference tabulates (update/UpdaterMSCKF.cpp:59-62)
 * msckf_<case>.npz   : inputs are regenerated from the seeded generator (ov_plane_amd/synth.py); the file stores the
                        CPU-restatement outputs (dx, P+, accept mask, chi2) of oracle/ovp_oracle.c for that scene.
 * wide_<what>.npz    : the same for the other rows of the path: plane loop, plane initialisation, SLAM update / delayed
                        initialisation, Propagator (Phi, Qd, mean), triangulation.
The reference itself cannot run in this image, so these are restatement outputs, not reference outputs.
"""
import os
import sys

import numpy as np
from scipy.stats import chi2

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle  # noqa: E402
from ov_plane_amd.synth import make_scene  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = {
    "sim11": dict(C=11, F=200, seed=0, chi2_mult=1.0),                    # BASELINE config[0] shape
    "ragged": dict(C=9, F=60, seed=2, ragged=True, chi2_mult=1.0),
    "nocalib": dict(C=6, F=40, seed=3, chi2_mult=1.0, calib=False),
    "nofej": dict(C=7, F=50, seed=4, chi2_mult=1.0, do_fej=False, ragged=True),
    "gate_all": dict(C=8, F=48, seed=5, chi2_mult=99999.0),               # sim config multiplier (bit-stable accept set)
    "c30": dict(C=30, F=120, seed=6, chi2_mult=1.0),                       # full window, reduced feature count
}

WIDE = {
    # plane-level gate wide open as in config/sim (chi2_multipler: 99999): the reference's plane statistic is rounding dependent
    "plane_loop": dict(C=9, F=120, seed=8, n_planes=6, feats_per_plane=12, chi2_mult=99999.0, ragged=True),
    "plane_init": dict(C=11, F=90, seed=33, n_planes=2, feats_per_plane=30, planes_in_state_frac=0.0, chi2_mult=1.0),
    "slam_update": dict(C=11, n_slam=14, seed=4, n_planes=3, outliers=2, wrong_plane=3),
    "slam_delayed": dict(C=11, F=8, seed=5, ragged=True),
    "triangulate": dict(C=11, F=200, seed=3, ragged=True, min_meas=2),
}


def wide_outputs(name):
    """Oracle outputs of the widened rows for the WIDE case `name` (dict of arrays)."""
    from ov_plane_amd.synth import PROP_OPTS, make_imu_scenario, make_slam_scene

    if name == "plane_loop":
        r = pyoracle.msckf_plane_update(make_scene(**WIDE[name]))
        return dict(P=r["P"], clone_p=r["clone_p"], clone_q=r["clone_q"], cp=r["cp"], used=r["used"], plane_ok=r["plane_ok"])
    if name == "plane_init":
        r = pyoracle.plane_init(make_scene(**WIDE[name]), const_init_multi=5.0, const_init_chi2=1.0)
        return dict(P=r["P"], clone_p=r["clone_p"], cp=r["cp"], used=r["used"], plane_ok=r["plane_ok"], new_id=r["new_id"])
    if name == "slam_update":
        sc = make_slam_scene(**WIDE[name])
        r = pyoracle.slam_update(sc, sc.lm_id, use_planes=True)
        return dict(P=r["P"], dx=r["dx"], accepted=r["accepted"], fellback=r["fellback"], chi2=r["chi2"])
    if name == "slam_delayed":
        r = pyoracle.slam_delayed_init(make_scene(**WIDE[name]))
        return dict(P=r["P"], ok=r["ok"], new_id=r["new_id"], p=r["p"], clone_p=r["clone_p"])
    if name == "triangulate":
        r = pyoracle.triangulate(make_scene(**WIDE[name]))
        return dict(p_FinG=r["p_FinG"], ok=r["ok"])
    if name == "propagate":
        out = {}
        for i, (rk4, fej) in enumerate([(1, 1), (0, 0)]):
            x, imu, t0, t1 = make_imu_scenario(7 + i)
            r = pyoracle.propagate_summed(x, dict(PROP_OPTS, use_rk4=rk4, do_fej=fej, imu_avg=0), imu, t0, t1)
            out["Phi%d" % i], out["Q%d" % i], out["last_w%d" % i] = r["Phi"], r["Q"], r["last_w"]
            out["x%d" % i] = np.concatenate([r["x"][k] for k in ("q", "p", "v", "bg", "ba")])
        return out
    raise KeyError(name)


WIDE_NAMES = list(WIDE) + ["propagate"]

if __name__ == "__main__":
    tab = np.zeros(1001)
    tab[1:] = chi2.ppf(0.95, np.arange(1, 1001))
    np.save(os.path.join(HERE, "chi2_095_table.npy"), tab)
    pyoracle.build()
    for name, kw in CASES.items():
        sc = make_scene(**kw)
        r = pyoracle.msckf_point_update(sc)
        np.savez_compressed(os.path.join(HERE, "msckf_%s.npz" % name), dx=r["dx"], P=r["P"], accepted=r["accepted"],
                            chi2=r["chi2"], rows_compressed=r["rows_compressed"])
        print(name, "accepted", int(r["accepted"].sum()), "/", sc.F)
    for name in WIDE_NAMES:
        np.savez_compressed(os.path.join(HERE, "wide_%s.npz" % name), **wide_outputs(name))
        print("wide", name)
    # a replayable frame in the binary trace format (ov_plane_amd/trace.py): inputs + the oracle's outputs
    from ov_plane_amd import trace

    sc = make_scene(seed=3, C=6, F=40, chi2_mult=1.0)
    trace.write_frames(os.path.join(HERE, "trace_c6.ovptrc"), [trace.frame_from_scene(sc, pyoracle.msckf_point_update(sc), 103.0)])
    print("trace_c6.ovptrc")
