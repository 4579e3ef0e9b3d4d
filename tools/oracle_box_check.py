#!/usr/bin/env python
"""Does the oracle give the same numbers on this machine as on the one that made tests/golden/plane_gate_ensemble.npz?
Runs every build of the oracle on a few fixture scenes and prints the largest difference per build (CPU only)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from oracle import pyoracle  # noqa: E402
from ov_plane_amd.synth import make_scene  # noqa: E402

z = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "plane_gate_ensemble.npz"))
pyoracle.build()
print(open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0])
for s in [int(a) for a in sys.argv[1:]] or [50, 61]:
    kw = json.loads(str(z["scenes"][s]))
    rows = np.where(z["scene"] == s)[0]
    sc = make_scene(**kw)
    ref = pyoracle.msckf_plane_update(sc)
    d = ref["plane_chi2"] - z["chi2"][rows, 0]
    print("scene", s, kw["F"], kw["seed"], "plain: decisions equal", bool((ref["plane_ok"] == z["ok"][rows]).all()), "max |d chi2|", float(np.abs(d).max()),
          "first differing plane", int(np.argmax(np.abs(d) > 0)) if (np.abs(d) > 0).any() else -1)
    for b, v in enumerate(("fma", "x87", "assoc"), start=1):
        so = pyoracle.build_variant(v)
        alt = pyoracle.msckf_plane_update(sc, libpath=so, force=z["ok"][rows])
        print("   ", v, float(np.abs(alt["plane_chi2"] - z["chi2"][rows, b]).max()))
