import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ov_plane_amd import capi
ctx = capi.Context(288, 30, 64)
rng = np.random.default_rng(0)
for n in [int(a) for a in sys.argv[1:]] or [240]:
    M = rng.standard_normal((n, n + 5)); A = M @ M.T / n + 0.1 * np.eye(n); b = rng.standard_normal(n)
    out = ctx.debug_chol2(A, b, add_identity=True, reps=50)
    print("n=%d dbg=%s: %.1f us" % (n, os.environ.get("OVP_C2_DBG", "0"), 1e3 * out["ms"]))
