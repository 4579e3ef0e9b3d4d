#!/usr/bin/env python
"""sha256 of every array of a generated scene (is ov_plane_amd/synth.py bit-reproducible across machines?)."""
import hashlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from ov_plane_amd.synth import make_scene  # noqa: E402

for kw in (dict(C=30, F=8000, seed=0, n_planes=50, feats_per_plane=50, planes_in_state_frac=0.5, chi2_mult=1.0),
           dict(C=30, F=2000, seed=11, n_planes=20, feats_per_plane=50, planes_in_state_frac=0.5, chi2_mult=1.0)):
    sc = make_scene(**kw)
    out = {}
    for k in sorted(sc.keys()):
        v = sc[k]
        if isinstance(v, np.ndarray):
            out[k] = hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest()[:12]
    print(json.dumps(out))
