"""Closed-loop visual-inertial odometry on simulated data: `sim.Simulator` feeds the C++ host mirror
(Propagator::propagate_and_clone -> UpdaterMSCKF::update with triangulation -> marginalize_old_clone, covariance and update on
the MI355X) and the estimated trajectory is scored against the simulator's ground truth (position / orientation error, NEES).

This is the glue `core/VioManager.cpp` provides around the update path, reduced to what a closed loop needs
(:348 propagate, :560-640 which tracks are used when, :670 MSCKF update, :864-866 marginalisation): a feature is used once its
track is lost or reaches back to the clone about to be marginalised, with every measurement still inside the window.
"""

import numpy as np

from .synth import radtan_undistort, state_layout


# ids up to 4 * max_aruco_features (1024 by default) belong to ArUco corners, which are never marginalised
# (state/StateHelper.cpp:638-652); ext TrackSIM shifts the simulator's map ids past them the same way
FID_OFFSET = 4 * 1024 + 1


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


def run_session(sim, n_frames=60, C=11, po=None, sigma_px=1.0, chi2_mult=1.0, planes=0, plane_min_feat=6, sigma_c=0.01,
                max_slam=0, feat_rep_slam=0, min_meas=3, out_dir=None, zupt=None, track_planes=True, forget_planes_at=None):
    """Closed loop through hostlib.Session, frame by frame, with the tracker-side bookkeeping of core/VioManager.cpp:360-506:
    a track is an MSCKF feature once it is lost or reaches back to the clone about to be marginalised; a track that spans the
    whole window (more than max_clone_size measurements) becomes a SLAM landmark while there is room (max_slam), and from then
    on every new measurement of it is a SLAM update until it is no longer seen (the session then marginalises it).
    feat_rep_slam: ext LandmarkRepresentation of the landmarks (0 GLOBAL_3D ... 5 ANCHORED_INVERSE_DEPTH_SINGLE).
    out_dir: write state_estimate.txt / state_deviation.txt / state_groundtruth.txt / timing.txt there, in the formats a run of
    the reference's simulation leaves behind (what its results/ scripts and ov_eval read).
    track_planes: hand the planes of all live tracks to the session every frame (the tracker's feature -> plane map,
    core/VioManager.cpp:513-534), so that a plane nobody observes any more leaves the state; forget_planes_at = frame index from
    which the tracker reports no plane at all (test hook: every plane of the state must then be marginalised).
    zupt: dict of Session.enable_zupt keywords (or {}) = VioManagerOptions::try_zupt: a frame at which the platform is found
    standing still gets a zero-velocity update instead of a clone (core/VioManager.cpp:311-331) and its measurements are dropped."""
    from . import hostlib
    from .sim import log_so3
    from .synth import PROP_OPTS, quat_2_rot

    po = dict(PROP_OPTS if po is None else po)
    po.update(sigma_w=sim.params["sigma_w"], sigma_a=sim.params["sigma_a"], sigma_wb=sim.params["sigma_wb"],
              sigma_ab=sim.params["sigma_ab"], gravity_mag=sim.params["gravity_mag"])
    imu, frames, plane_of = collect(sim, C + 1 + n_frames)
    init = initial_state(sim, frames, C)
    frames = frames[:C] + frames[C + 1:]
    ses = hostlib.Session(init, po, sigma_px=sigma_px, chi2_mult=chi2_mult, plane_mode=planes, plane_min_feat=plane_min_feat,
                          sigma_c=sigma_c, max_slam=max_slam, feat_rep_slam=feat_rep_slam, cam_dt=1.0 / sim.params["sim_freq_cam"])
    if zupt is not None:
        ses.enable_zupt(po, **zupt)
    ses.feed_imu(imu)
    if out_dir is not None:
        import os

        os.makedirs(out_dir, exist_ok=True)
        ses.open_files(*(os.path.join(out_dir, n) for n in ("state_estimate.txt", "state_deviation.txt", "state_groundtruth.txt",
                                                            "timing.txt")))
    from .sim import P_IINC, R_ITOC
    from .synth import rot_2_quat

    truth_tail = np.concatenate([[sim.params["calib_camimu_dt"]], sim.intr, rot_2_quat(R_ITOC), P_IINC])
    tracks = {}
    for k in range(C):
        for fid, uv in frames[k][1].items():
            tracks.setdefault(fid, []).append((k, uv))
    window = list(range(C))                                   # frames that have a clone in the state, oldest first
    if zupt is not None:
        fk = frames[C - 1]
        ses.feed_tracks(fk[0], np.array(list(fk[1]), dtype=np.int64) + FID_OFFSET, np.array(list(fk[1].values())).reshape(-1, 2))
    slam_ids = set()
    K = len(frames) - C
    traj, posecov, counts = np.zeros((K, 16)), np.zeros((K, 6, 6)), np.zeros((K, 6), dtype=np.int32)
    n_feats = np.zeros((K, 3), dtype=np.int32)
    zupt_frames = np.zeros(K, dtype=bool)
    for k in range(C, len(frames)):
        t_k, seen = frames[k]
        i = k - C
        if zupt is not None:
            ses.feed_tracks(t_k, np.array(list(seen), dtype=np.int64) + FID_OFFSET, np.array(list(seen.values())).reshape(-1, 2))
            z = ses.try_zupt(t_k)
            if z is not None:                                 # standing still: no clone, this image's measurements are dropped
                traj[i], posecov[i], zupt_frames[i] = z["x16"], z["posecov"], True
                counts[i, 4], counts[i, 5] = len(slam_ids), counts[i - 1, 5] if i else 0
                continue
        win = window + [k]
        in_win = {fr: j for j, fr in enumerate(win)}
        lo = win[0]
        items = []                                           # (fid, kind, [(frame, uv), ...])
        for fid in sorted(slam_ids):                         # landmarks seen in this frame: one new measurement each
            if fid in seen:
                items.append((fid, 1, [(k, seen[fid])]))
        n_slam_kept = len(items)
        for fid, uv in seen.items():
            if fid not in slam_ids:
                tracks.setdefault(fid, []).append((k, uv))
        maxtracks, msckf = [], []
        for fid in sorted(tracks):
            tr = [(j, uv) for j, uv in tracks[fid] if j in in_win]
            tracks[fid] = tr
            if not tr:
                del tracks[fid]
                continue
            lost, marg = fid not in seen, tr[0][0] == lo
            if lost or marg:
                if not lost and len(tr) > C:                 # :418-436 reached the maximum track length
                    maxtracks.append((fid, tr))
                elif len(tr) >= min_meas:
                    msckf.append((fid, tr))
                del tracks[fid]
        room = max(max_slam - n_slam_kept, 0)                # :449-460 the last ones of the list become landmarks
        new_slam = maxtracks[len(maxtracks) - min(room, len(maxtracks)):] if room else []
        rest = maxtracks[:len(maxtracks) - len(new_slam)]
        items += [(fid, 2, tr) for fid, tr in new_slam] + [(fid, 0, tr) for fid, tr in msckf + rest]
        F, M = len(items), C + 1
        uv = np.zeros((F, M, 2), dtype=np.float32)
        slot = -np.ones((F, M), dtype=np.int32)
        nm = np.zeros(F, dtype=np.int32)
        for f, (_, _, tr) in enumerate(items):
            nm[f] = len(tr)
            for q, (j, m) in enumerate(tr):
                uv[f, q], slot[f, q] = m, in_win[j]
        xn, yn = radtan_undistort(uv[..., 0], uv[..., 1], init["intr"])
        uvn = np.stack([xn, yn], axis=-1).astype(np.float32)
        fid = np.array([it[0] for it in items], dtype=np.int64)
        kind = np.array([it[1] for it in items], dtype=np.int32)
        pl = np.array([max(plane_of[int(f)], 0) for f in fid], dtype=np.int32) if planes else None
        gt = sim.get_state(t_k + sim.params["calib_camimu_dt"])
        truth = np.concatenate([[gt["t"]], gt["q"], gt["p"], gt["v"], gt["bg"], gt["ba"], truth_tail])
        active = None
        if planes == 2 and track_planes:
            live = set(seen) | set(tracks) | {it[0] for it in items}
            active = {plane_of[int(f)] for f in live if plane_of[int(f)] > 0}
            if forget_planes_at is not None and i >= forget_planes_at:
                active, pl = set(), np.zeros(len(fid), dtype=np.int32)
        out = ses.step(t_k, uv, uvn, slot, nm, fid + FID_OFFSET, kind, pl, truth, active_planes=active)
        window = win[1:]
        slam_ids = {i_ - FID_OFFSET for i_ in out["slam_ids"]}
        for f in slam_ids:
            tracks.pop(f, None)
        traj[i], posecov[i], counts[i] = out["x16"], out["posecov"], out["counts"]
        n_feats[i] = [(kind == 0).sum(), (kind == 1).sum(), (kind == 2).sum()]
    ses.close()
    times = np.array([t for t, _ in frames[C:]])
    e_pos, e_ori, nees_p, nees_o = np.zeros(K), np.zeros(K), np.zeros(K), np.zeros(K)
    for i in range(K):
        gt = sim.get_state(times[i] + sim.params["calib_camimu_dt"])
        dp = traj[i, 4:7] - gt["p"]
        dth = log_so3(quat_2_rot(traj[i, 0:4]) @ quat_2_rot(gt["q"]).T)
        e_pos[i], e_ori[i] = np.linalg.norm(dp), np.linalg.norm(dth)
        nees_p[i] = dp @ np.linalg.solve(posecov[i][3:6, 3:6], dp)
        nees_o[i] = dth @ np.linalg.solve(posecov[i][0:3, 0:3], dth)
    return dict(times=times, traj=traj, posecov=posecov, counts=counts, n_feats=n_feats, zupt_frames=zupt_frames, e_pos=e_pos,
                e_ori=e_ori, nees_pos=nees_p, nees_ori=nees_o, rmse_pos=float(np.sqrt(np.mean(e_pos**2))),
                rmse_ori_deg=float(np.degrees(np.sqrt(np.mean(e_ori**2)))))
