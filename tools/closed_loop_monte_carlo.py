"""Closed-loop Monte-Carlo of the filter session on the simulator: python tools/closed_loop_monte_carlo.py [n_seeds]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from ov_plane_amd.sim import Simulator, synthetic_trajectory
from ov_plane_amd import closed_loop, hostlib

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
np.set_printoptions(precision=3, suppress=True, linewidth=220)
traj_syn = synthetic_trajectory(duration=30.0)
traj_udel = hostlib.load_trajectory(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "udel_arl_short_60s.txt"))
for name, traj, nf in (("synthetic loop", traj_syn, 150), ("udel_arl_short head", traj_udel, 300)):
    for (planes, max_slam) in ((0, 0), (0, 25), (2, 25)):
        rows = []
        for seed in range(n_seeds):
            sim = Simulator(traj, num_pts=100, num_pts_plane=100, sim_seed_measurements=seed, sim_seed_state_init=seed)
            r = closed_loop.run_session(sim, n_frames=nf, C=11, planes=planes, max_slam=max_slam)
            rows.append([r["rmse_pos"], r["rmse_ori_deg"], r["nees_pos"].mean(), r["nees_ori"].mean(), r["e_pos"][-1]])
        m = np.array(rows).mean(axis=0)
        print("%-20s planes=%d slam=%2d | rmse pos %.3f m ori %.3f deg | nees pos %.2f ori %.2f | final drift %.3f m" % (name, planes, max_slam, *m))
