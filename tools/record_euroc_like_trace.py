#!/usr/bin/env python
"""Records the per-frame trace of a closed-loop run at the sizes of the reference's real-data configuration
(config/euroc_mav/estimator_config.yaml:16-19,155: max_clones 11, at most 20 MSCKF features per update, chi2_multipler 1):
simulated data (ov_plane_amd/sim.py) through propagate -> triangulate -> UpdaterMSCKF::update -> marginalise on the device, every
point update dumped with its inputs and outputs in the OVPTRC01 format (ov_plane_amd/trace.py).  The stand-in for BASELINE
config 5 (EuRoC replay through ROS, not runnable here): the frames can be replayed against the oracle (tests) or next to the
reference wherever ROS + open_vins exist.  Usage: python tools/record_euroc_like_trace.py OUT.ovptrc [n_frames] [keep_every]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    out = sys.argv[1]
    n_frames = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    keep_every = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    from ov_plane_amd import closed_loop, hostlib, trace
    from ov_plane_amd.build import build_host
    from ov_plane_amd.sim import Simulator, synthetic_trajectory

    build_host()
    sim = Simulator(synthetic_trajectory(duration=30.0), num_pts=150, num_pts_plane=0)
    tmp = out + ".all"
    assert hostlib.lib().ovph_update_trace(tmp.encode()) == 0
    r = closed_loop.run(sim, n_frames=n_frames, C=11, chi2_mult=1.0, max_feats=20)
    hostlib.lib().ovph_update_trace(None)
    frames = trace.read_frames(tmp)
    os.remove(tmp)
    kept = frames[::keep_every]
    trace.write_frames(out, kept)
    print("recorded %d updates, kept %d; features per update %s; rmse %.3f m" % (
        len(frames), len(kept), sorted({int(f["F"]) for f in frames}), r["rmse_pos"]))


if __name__ == "__main__":
    main()
