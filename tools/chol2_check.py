#!/usr/bin/env python
"""k_chol2 against numpy on random SPD matrices (factor, z = L^-1 b, y = L^-T z) and its launch time; old k_tilechol for comparison."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from ov_plane_amd import capi  # noqa: E402


def main():
    ctx = capi.Context(288, 30, 64)
    rng = np.random.default_rng(0)
    for n in [5, 16, 31, 48, 96, 100, 197, 210, 240, 255, 256, 271, 272, 285, 287]:
        for border in [False, True]:
            M = rng.standard_normal((n, n + 5))
            A = M @ M.T / n + 0.1 * np.eye(n)
            b = rng.standard_normal(n) if border else None
            out = ctx.debug_chol2(A, b, add_identity=True, reps=20)
            Lr = np.linalg.cholesky(A + np.eye(n))
            L = out["L"][:n, :n]
            eL = np.abs(L - Lr).max()
            msg = "n=%3d border=%d rc=%d |dL| %.2e" % (n, border, out["rc"], eL)
            if border:
                z = np.linalg.solve(Lr, b)
                y = np.linalg.solve(Lr.T, z)
                msg += " |dz| %.2e |dy| %.2e |brow-row| %.2e" % (np.abs(out["z"] - z).max(), np.abs(out["y"] - y).max(),
                                                                 np.abs(out["L"][n, :n] - z).max())
            piv = np.diag(Lr) ** 2
            msg += " |dpiv| %.2e  %.1f us" % (np.abs(out["piv"] - piv).max(), 1e3 * out["ms"])
            print(msg, flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
