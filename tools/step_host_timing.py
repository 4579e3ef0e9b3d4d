#!/usr/bin/env python
"""Host clock of the five C-ABI calls of the config-3 step (perf_counter around each call, 300 steps after 100 warm-up steps) and the
step time with / without the batch upload.  GPU box only."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import bench  # noqa: E402


def main():
    import torch

    be = bench.load_backend(torch, None)
    be.set_device(0)
    sc = be.make_workload("config3")
    run = be.make_runner(sc, 0)
    fr, lib, ctx = run.frame, run._lib, run.ctx
    import ctypes as C
    names = ["cov_set_device", "state_upload", "batch_upload", "plane_update", "point_update"]
    with be.stream_ctx(run):
        for _ in range(100):
            run.step()
        acc = np.zeros(5)
        K = 300
        t_all0 = time.perf_counter()
        for _ in range(K):
            t0 = time.perf_counter()
            lib.ovp_cov_set_device(ctx._h, run._P0_ptr, sc.N, sc.N)
            t1 = time.perf_counter()
            lib.ovp_state_upload(ctx._h, C.byref(fr.st))
            t2 = time.perf_counter()
            lib.ovp_batch_upload(ctx._h, C.byref(fr.fb))
            t3 = time.perf_counter()
            fr.plane_update()
            t4 = time.perf_counter()
            fr.point_update()
            t5 = time.perf_counter()
            acc += np.diff([t0, t1, t2, t3, t4, t5])
        t_all = (time.perf_counter() - t_all0) / K
        print("step %.1f us; per call (us): %s" % (1e6 * t_all, ", ".join("%s %.1f" % (n, 1e6 * a / K) for n, a in zip(names, acc))))
        for label, fn in (("step", run.step), ("step_resident", run.step_resident), ("step", run.step), ("step_resident", run.step_resident)):
            for _ in range(20):
                fn()
            t0 = time.perf_counter()
            for _ in range(K):
                fn()
            print("%s: %.1f us per step" % (label, 1e6 * (time.perf_counter() - t0) / K))


if __name__ == "__main__":
    main()
