"""Generates the committed fixtures under tests/golden/ (run from the repo root: python tests/golden/make_golden.py).

 * chi2_095_table.npy : scipy.stats.chi2.ppf(0.95, k), k = 0..1000 (k=0 -> 0), the pin for the boost quantile the
                        reference tabulates (update/UpdaterMSCKF.cpp:59-62)
 * msckf_<case>.npz   : inputs are regenerated from the seeded generator (ov_plane_amd/synth.py); the file stores the
                        CPU-restatement outputs (dx, P+, accept mask, chi2) of oracle/ovp_oracle.c for that scene.
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
