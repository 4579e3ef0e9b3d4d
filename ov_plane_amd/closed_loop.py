"""Closed-loop visual-inertial odometry on simulated data: `sim.Simulator` feeds the C++ host mirror
(Propagator::propagate_and_clone -> UpdaterMSCKF::update with triangulation -> marginalize_old_clone, covariance and update on
the MI355X) and the estimated trajectory is scored against the simulator's ground truth (position / orientation error, NEES).

This is the glue `core/VioManager.cpp` provides around the update path, reduced to what a closed loop needs
(:348 propagate, :560-640 which tracks are used when, :670 MSCKF update, :864-866 marginalisation): a feature is used once its
track is lost or reaches back to the clone about to be marginalised, with every measurement still inside the window.
"""

import numpy as np

from .synth import radtan_undistort, state_layout


def collect(sim, n_frames):
    """Runs the simulator until n_frames camera frames exist.  Returns (imu [n,7], frames [(time_cam, {fid: uv})],
    plane_of {fid: plane id or -1})."""
    imu, frames, plane_of = [], [], {}
    while sim.is_running and len(frames) < n_frames:
        r = sim.get_next_imu()
        if r is not None:
            imu.append(np.concatenate([[r[0]], r[1], r[2]]))
        c = sim.get_next_cam()
        if c is not None:
            frames.append((c[0], {fid: d[:2].copy() for fid, d in c[1]}))
            plane_of.update({fid: int(d[2]) for fid, d in c[1]})
    # readings past the last frame so the final propagation has its bounding measurement
    for _ in range(4):
        r = sim.get_next_imu()
        if r is not None:
            imu.append(np.concatenate([[r[0]], r[1], r[2]]))
    return np.array(imu), frames, plane_of


def initial_state(sim, frames, C, rng=None, sigmas=None):
    """Filter state at the time of frame C: a window of C clones (frames 0..C-1) and the IMU (at the time of frame C, which
    is not cloned) at ground truth, with the prior of state/State.cpp:85-101 on the calibration and a small uncorrelated prior
    on the poses."""
    ids = state_layout(C)
    N = ids["N"]
    sg = dict(ori=1e-3, pos=1e-2, vel=1e-2, bg=1e-4, ba=1e-3, dt=1e-3, cal_ori=1e-3, cal_pos=1e-2, focal=1.0, dist=5e-3)
    sg.update(sigmas or {})
    d = np.zeros(N)
    d[0:3], d[3:6], d[6:9], d[9:12], d[12:15] = sg["ori"], sg["pos"], sg["vel"], sg["bg"], sg["ba"]
    d[15] = sg["dt"]
    d[16:19], d[19:22] = sg["cal_ori"], sg["cal_pos"]
    d[22:26], d[26:30] = sg["focal"], sg["dist"]
    for c in range(C):
        d[30 + 6 * c:33 + 6 * c], d[33 + 6 * c:36 + 6 * c] = sg["ori"], sg["pos"]
    cq, cp = np.zeros((C, 4)), np.zeros((C, 3))
    for c in range(C):
        st = sim.get_state(frames[c][0] + sim.params["calib_camimu_dt"])
        cq[c], cp[c] = st["q"], st["p"]
    st = sim.get_state(frames[C][0] + sim.params["calib_camimu_dt"])
    x = dict(q=st["q"], p=st["p"], v=st["v"], bg=st["bg"], ba=st["ba"])
    x.update({k + "_fej": v.copy() for k, v in list(x.items())})
    from .sim import P_IINC, R_ITOC
    from .synth import rot_2_quat

    return dict(C=C, N=N, clone_q=cq, clone_p=cp, clone_q_fej=cq.copy(), clone_p_fej=cp.copy(), calib_q=rot_2_quat(R_ITOC),
                calib_p=P_IINC.copy(), intr=sim.intr.copy(), x=x, dt=float(sim.params["calib_camimu_dt"]), P=np.diag(d * d),
                t_state=frames[C][0])


def schedule(frames, C, min_meas=3, max_feats=None):
    """For every live frame k >= C: the tracks used at that frame (lost in frame k, or reaching back to the clone that is
    marginalised after this update) with the window slots of their measurements.  Tracks restart after use, as the reference's
    feature database does when a used feature is deleted and its id shows up again."""
    tracks = {}
    for k in range(C):
        for fid, uv in frames[k][1].items():
            tracks.setdefault(fid, []).append((k, uv))
    out = []
    for k in range(C, len(frames)):
        seen = frames[k][1]
        for fid, uv in seen.items():
            tracks.setdefault(fid, []).append((k, uv))
        lo = k - C                                    # oldest clone of the window (C + 1 clones: lo .. k)
        use = []
        for fid in sorted(tracks):
            tr = [(j, uv) for j, uv in tracks[fid] if j >= lo]
            tracks[fid] = tr
            if not tr:
                del tracks[fid]
                continue
            lost = fid not in seen
            marg = tr[0][0] == lo
            if lost or marg:
                if len(tr) >= min_meas:
                    use.append((fid, tr))
                del tracks[fid]
        if max_feats is not None:
            use = sorted(use, key=lambda e: len(e[1]), reverse=True)[:max_feats]
        F, M = len(use), C + 1
        uv = np.zeros((F, M, 2), dtype=np.float32)
        slot = -np.ones((F, M), dtype=np.int32)
        nm = np.zeros(F, dtype=np.int32)
        for f, (_, tr) in enumerate(use):
            nm[f] = len(tr)
            for q, (j, m) in enumerate(tr):
                uv[f, q] = m
                slot[f, q] = j - lo
        out.append(dict(uv=uv, slot=slot, n_meas=nm, fid=np.array([fid for fid, _ in use], dtype=np.int64)))
    return out


def run(sim, n_frames=60, C=11, po=None, sigma_px=1.0, chi2_mult=1.0, max_feats=None, planes=0, plane_min_feat=6, sigma_c=0.01):
    """Simulate, estimate, score.  Returns dict with per-frame truth / estimate / errors and the summary numbers.
    planes: 0 = points only; 1 = the simulator's point-to-plane associations become feat2plane and UpdaterMSCKF uses the
    planar regularities (planes estimated per update, StateOptions::use_plane_constraint_msckf); 2 = UpdaterPlane::init_vio_plane
    also puts the planes into the state (use_plane_slam_feats)."""
    from . import hostlib
    from .sim import log_so3
    from .synth import PROP_OPTS, quat_2_rot

    po = dict(PROP_OPTS if po is None else po)
    po.update(sigma_w=sim.params["sigma_w"], sigma_a=sim.params["sigma_a"], sigma_wb=sim.params["sigma_wb"],
              sigma_ab=sim.params["sigma_ab"], gravity_mag=sim.params["gravity_mag"])
    imu, frames, plane_of = collect(sim, C + 1 + n_frames)
    init = initial_state(sim, frames, C)
    frames = frames[:C] + frames[C + 1:]      # the image taken at the initial state time has no clone
    live = schedule(frames, C, max_feats=max_feats)
    for fr in live:   # what the tracker stores next to the raw pixels (ext TrackSIM: undistort with the calibration estimate)
        xn, yn = radtan_undistort(fr["uv"][..., 0], fr["uv"][..., 1], init["intr"])
        fr["uv_norm"] = np.stack([xn, yn], axis=-1).astype(np.float32)
    times = np.array([t for t, _ in frames[C:]])
    if planes:
        for fr in live:
            fr["plane"] = np.array([max(plane_of[int(f)], 0) for f in fr["fid"]], dtype=np.int32)
    out = hostlib.run_sequence(init, imu, times, live, po, sigma_px=sigma_px, chi2_mult=chi2_mult, trace=True, plane_mode=planes,
                               plane_min_feat=plane_min_feat, sigma_c=sigma_c)
    K = len(live)
    e_pos, e_ori, nees_p, nees_o = np.zeros(K), np.zeros(K), np.zeros(K), np.zeros(K)
    for k in range(K):
        gt = sim.get_state(times[k] + sim.params["calib_camimu_dt"])
        q, p = out["traj"][k, 0:4], out["traj"][k, 4:7]
        dp = p - gt["p"]
        dth = log_so3(quat_2_rot(q) @ quat_2_rot(gt["q"]).T)
        Pk = out["posecov"][k]
        e_pos[k], e_ori[k] = np.linalg.norm(dp), np.linalg.norm(dth)
        nees_p[k] = dp @ np.linalg.solve(Pk[3:6, 3:6], dp)
        nees_o[k] = dth @ np.linalg.solve(Pk[0:3, 0:3], dth)
    return dict(times=times, traj=out["traj"], posecov=out["posecov"], e_pos=e_pos, e_ori=e_ori, nees_pos=nees_p, nees_ori=nees_o,
                feats_per_frame=np.array([len(fr["n_meas"]) for fr in live]), kept_per_frame=out["kept"],
                planar_per_frame=np.array([int((fr["plane"] > 0).sum()) if planes else 0 for fr in live]),
                planes_in_state=out.get("planes_in_state", 0),
                rmse_pos=float(np.sqrt(np.mean(e_pos**2))), rmse_ori_deg=float(np.degrees(np.sqrt(np.mean(e_ori**2)))),
                final=out)
