"""ctypes access to libovplane_host.so: the C++ host mirror of ov_plane's State / StateHelper / UpdaterMSCKF
(ov_plane_amd/csrc/host/).  Only a harness symbol is exported; the classes themselves are C++."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libovplane_host.so")
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libovplane_host.so is not built (run __graft_entry__.build())")
        C.CDLL(os.path.join(_HERE, "libovplane_hip.so"), mode=C.RTLD_GLOBAL)
        _LIB = C.CDLL(LIB_PATH)
    return _LIB


def run_msckf_update(sc, triangulate=False, fit_planes=None, comm=None, rank=0, world=1, device=0):
    """Drives ov_plane::UpdaterMSCKF::update (C++ host classes over the C-ABI) on a synth.Scene.
    triangulate=True: the features carry uvs_norm and no position; the updater triangulates them first.
    fit_planes=dict(min_feat, max_cond, variant): no plane estimates are handed over - the updater fits the planes that are
    not in the state (PlaneFitting::plane_fitting) and refines planes and on-plane features (optimize_plane) itself.
    comm / rank / world / device: UpdaterMSCKF::set_communicator + StateOptions::gpu_device - the point loop goes through
    ovp_msckf_update_sharded on this rank's share (out["shard"] = its index range of the point batch)."""
    L = lib()
    cam1 = sc.get("cam1", None)  # synth.make_stereo_scene: dict(calib_q, calib_p, intr), sc.cam_idx [F, M]
    if cam1 is not None:
        c1q, c1p, c1i = (np.ascontiguousarray(cam1[k], dtype=np.float64) for k in ("calib_q", "calib_p", "intr"))
        cam_of = np.ascontiguousarray(sc.cam_idx, dtype=np.int32)
        L.ovph_set_second_camera(c1q.ctypes.data_as(C.c_void_p), c1p.ctypes.data_as(C.c_void_p), c1i.ctypes.data_as(C.c_void_p),
                                 cam_of.ctypes.data_as(C.c_void_p))
    L.ovph_set_shard_comm.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.ovph_set_shard_comm(C.c_void_p(comm) if comm else None, int(rank), int(world), int(device))
    L.ovph_set_fisheye(1 if sc.get("fisheye", False) else 0)
    if fit_planes is not None:
        L.ovph_set_plane_fit.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int]
        L.ovph_set_plane_fit(1, int(fit_planes["min_feat"]), float(fit_planes["max_cond"]), int(fit_planes.get("variant", 0)))
    uvn = np.ascontiguousarray(sc.uv_norm, dtype=np.float32) if triangulate else None
    L.ovph_set_uv_norm(uvn.ctypes.data_as(C.c_void_p) if triangulate else None)
    f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)  # noqa: E731
    in_state = np.where(sc.plane_in_state)[0]
    out_state = np.where(~sc.plane_in_state)[0]
    ids_in = np.ascontiguousarray(in_state + 1, dtype=np.uint64)
    ids_out = np.ascontiguousarray(out_state + 1, dtype=np.uint64)
    cp_in, cpf_in, cp_out = f64(sc.cp[in_state]), f64(sc.cp_fej[in_state]), f64(sc.cp[out_state])
    N, F, M = int(sc.N), int(sc.F), int(sc.uv.shape[1])
    P = np.asfortranarray(sc.P)
    uv = np.ascontiguousarray(sc.uv, dtype=np.float32)
    cidx = np.ascontiguousarray(sc.clone_idx, dtype=np.int32)
    nm = np.ascontiguousarray(sc.n_meas, dtype=np.int32)
    pf = f64(sc.p_FinG)
    pof = np.ascontiguousarray(sc.plane_id, dtype=np.int32)
    o = sc.opts
    out = dict(clone_q=np.zeros((sc.C, 4)), clone_p=np.zeros((sc.C, 3)), calib_q=np.zeros(4), calib_p=np.zeros(3),
               intr=np.zeros(8), cp_state=np.zeros((max(len(in_state), 1), 3)), P=np.zeros((N, N)),
               kept=np.zeros(F, dtype=np.uint8), used=np.zeros(F, dtype=np.uint8), deleted=np.zeros(F, dtype=np.uint8))
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    cq, cp_, cqf, cpf = f64(sc.clone_q), f64(sc.clone_p), f64(sc.clone_q_fej), f64(sc.clone_p_fej)
    calq, calp, intr = f64(sc.calib_q), f64(sc.calib_p), f64(sc.intr)
    L.ovph_run_msckf_update.restype = C.c_int
    rc = L.ovph_run_msckf_update(
        C.c_int(sc.C), p(cq), p(cp_), p(cqf), p(cpf), p(calq), p(calp), p(intr), C.c_int(len(in_state)), p(cp_in), p(cpf_in),
        p(ids_in), C.c_int(len(out_state)), p(cp_out), p(ids_out), C.c_int(N), p(P), C.c_int(F), C.c_int(M), p(uv), p(cidx),
        p(nm), p(pf), p(pof), C.c_double(o["sigma_px"]), C.c_double(o["chi2_mult"]), C.c_double(o["sigma_c"]),
        C.c_int(int(o["do_fej"])), p(out["clone_q"]), p(out["clone_p"]), p(out["calib_q"]), p(out["calib_p"]), p(out["intr"]),
        p(out["cp_state"]), p(out["P"]), p(out["kept"]), p(out["used"]), p(out["deleted"]))
    if rc != 0:
        raise RuntimeError("ovph_run_msckf_update failed with %d" % rc)
    out["cp_state"] = out["cp_state"][: len(in_state)]
    for k in ("kept", "used", "deleted"):
        out[k] = out[k].astype(bool)
    lo, hi = C.c_int(0), C.c_int(0)
    L.ovph_last_shard(C.byref(lo), C.byref(hi))
    out["shard"] = (lo.value, hi.value)
    if cam1 is not None:
        c1 = np.zeros(15)
        L.ovph_last_second_camera(c1.ctypes.data_as(C.c_void_p))
        out["cam1"] = dict(calib_q=c1[:4].copy(), calib_p=c1[4:7].copy(), intr=c1[7:].copy())
    return out


def run_initialize(sc, order, H_R, H_L, res, r_iso, chi2_mult, new_value0):
    """Drives ov_plane::StateHelper::initialize (C++ host mirror) on the state of a synth.Scene (no planes / SLAM)."""
    L = lib()
    f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)  # noqa: E731
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    N = int(sc.N)
    rows, k = H_L.shape
    cols = H_R.shape[1]
    P = np.asfortranarray(sc.P)
    HR = np.asfortranarray(H_R, dtype=np.float64)
    HL = np.asfortranarray(H_L, dtype=np.float64)
    r = f64(res)
    oid = np.ascontiguousarray([o[0] for o in order], dtype=np.int32)
    cq, cp_ = f64(sc.clone_q), f64(sc.clone_p)
    v0 = f64(new_value0)
    out = dict(P=np.zeros((N + k, N + k)), new_value=np.zeros(k), clone_q=np.zeros((sc.C, 4)), clone_p=np.zeros((sc.C, 3)),
               calib_p=np.zeros(3), intr=np.zeros(8))
    L.ovph_run_initialize.restype = C.c_int
    rc = L.ovph_run_initialize(C.c_int(sc.C), p(cq), p(cp_), C.c_int(N), p(P), C.c_int(len(order)), p(oid), C.c_int(rows),
                               C.c_int(cols), p(HR), C.c_int(k), p(HL), p(r), C.c_double(r_iso), C.c_double(chi2_mult), p(v0),
                               p(out["P"]), p(out["new_value"]), p(out["clone_q"]), p(out["clone_p"]), p(out["calib_p"]),
                               p(out["intr"]))
    if rc < 0:
        raise RuntimeError("ovph_run_initialize failed with %d" % rc)
    out["ok"] = rc
    return out


def set_slam_force_dense(on: bool):
    """UpdaterSLAM::update through its dense form (the fallback of batches the device entry refuses) until switched off."""
    lib().ovph_set_slam_force_dense(C.c_int(1 if on else 0))


def run_updater(sc, mode, const_init_multi=5.0, const_init_chi2=1.0, fit_planes=None, slam=None, slam_rep=None,
                feat_rep_slam=None):
    """Drives the C++ host mirrors on a synth scene.
    fit_planes=dict(min_feat, max_cond, variant) (mode "plane_init" only): the features carry normalised measurements and
    no position, no plane estimates are handed over - init_vio_plane triangulates, fits and refines itself.

    mode "slam_update": UpdaterSLAM::update on a synth.make_slam_scene (feature f observes landmark f);
    mode "slam_delayed_init": UpdaterSLAM::delayed_init on a plain scene (every feature is a landmark candidate);
    mode "plane_init": UpdaterPlane::init_vio_plane on a scene whose planes are all out of the state.
    """
    L = lib()
    L.ovph_set_fisheye(1 if sc.get("fisheye", False) else 0)
    if slam_rep is not None:  # mode "slam_update": (representation, anchor clone slot) of every landmark; out["slam_p"] are its parameters
        L.ovph_set_slam_rep(int(slam_rep[0]), int(slam_rep[1]))
    if feat_rep_slam is not None:  # mode "slam_delayed_init": StateOptions::feat_rep_slam; out["new_p"] are landmark parameters
        L.ovph_set_feat_rep_slam(int(feat_rep_slam))
    # "msckf_fit": UpdaterMSCKF::update on a state with SLAM landmarks (slam = dict(p [k,3], p_fej [k,3], plane [k]); the
    # scene must have n_slam = k landmark columns) and the scene's in-state planes; needs fit_planes
    m = {"slam_update": 0, "slam_delayed_init": 1, "plane_init": 2, "msckf_fit": 3}[mode]
    uvn_keep = None
    if fit_planes is not None:
        assert m in (2, 3)
        uvn_keep = np.ascontiguousarray(sc.uv_norm, dtype=np.float32)
        L.ovph_set_uv_norm(uvn_keep.ctypes.data_as(C.c_void_p))
        L.ovph_set_plane_fit.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int]
        L.ovph_set_plane_fit(1, int(fit_planes["min_feat"]), float(fit_planes["max_cond"]), int(fit_planes.get("variant", 0)))
    f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)  # noqa: E731
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    N, F, M = int(sc.N), int(sc.F), int(sc.uv.shape[1])
    n_slam = F if m == 0 else 0
    n_pl_total = int(sc.cp.shape[0])
    n_pl_in = n_pl_total if m == 0 else 0
    n_pl_out = n_pl_total if m == 2 else 0
    n_cap = N + 3 * F + 3 * n_pl_total + 8
    dummy = np.zeros((1, 3))
    slam_p = f64(sc.slam_p if n_slam else dummy)
    slam_pf = f64(sc["p_FinG_fej"] if n_slam else dummy)
    slam_pl_keep = None
    if m == 3:
        assert fit_planes is not None and slam is not None
        n_slam = len(slam["plane"])
        n_pl_in = int(np.sum(sc.plane_in_state))
        assert sc.plane_in_state[:n_pl_in].all()
        slam_p, slam_pf = f64(slam["p"]), f64(slam["p_fej"])
        slam_pl_keep = np.ascontiguousarray(slam["plane"], dtype=np.int32)
        L.ovph_set_slam_planes(slam_pl_keep.ctypes.data_as(C.c_void_p))
    cp = f64(sc.cp if n_pl_total else dummy)
    cpf = f64(sc.cp_fej if n_pl_total else dummy)
    P = np.asfortranarray(sc.P)
    uv = np.ascontiguousarray(sc.uv, dtype=np.float32)
    cidx = np.ascontiguousarray(sc.clone_idx, dtype=np.int32)
    nm = np.ascontiguousarray(sc.n_meas, dtype=np.int32)
    pf = f64(sc.p_FinG)
    pof = np.ascontiguousarray(sc.plane_id, dtype=np.int32)
    o = sc.opts
    nout = max(F, n_pl_total, 1)
    out = dict(clone_q=np.zeros((sc.C, 4)), clone_p=np.zeros((sc.C, 3)), calib_q=np.zeros(4), calib_p=np.zeros(3),
               intr=np.zeros(8), slam_p=np.zeros((max(n_slam, 1), 3)), cp=np.zeros((max(n_pl_in, 1), 3)),
               P=np.zeros((n_cap, n_cap)), n=np.zeros(1, dtype=np.int32), kept=np.zeros(F, dtype=np.uint8),
               deleted=np.zeros(F, dtype=np.uint8), should_marg=np.zeros(max(n_slam, 1), dtype=np.uint8),
               slam_to_plane=np.zeros(F, dtype=np.int32), new_p=np.zeros((nout, 3)), new_id=np.zeros(nout, dtype=np.int32))
    cq, cp_, cqf, cpf_ = f64(sc.clone_q), f64(sc.clone_p), f64(sc.clone_q_fej), f64(sc.clone_p_fej)
    calq, calp, intr = f64(sc.calib_q), f64(sc.calib_p), f64(sc.intr)
    L.ovph_run_updater.restype = C.c_int
    rc = L.ovph_run_updater(
        C.c_int(m), C.c_int(sc.C), p(cq), p(cp_), p(cqf), p(cpf_), p(calq), p(calp), p(intr), C.c_int(n_slam), p(slam_p),
        p(slam_pf), C.c_int(n_pl_in), p(cp), p(cpf), C.c_int(n_pl_out), p(cp), C.c_int(N), p(P), C.c_int(F), C.c_int(M), p(uv),
        p(cidx), p(nm), p(pf), p(pof), C.c_double(o["sigma_px"]), C.c_double(o["chi2_mult"]), C.c_double(o["sigma_c"]),
        C.c_int(int(o["do_fej"])), C.c_double(const_init_multi), C.c_double(const_init_chi2), C.c_int(n_cap),
        p(out["clone_q"]), p(out["clone_p"]), p(out["calib_q"]), p(out["calib_p"]), p(out["intr"]), p(out["slam_p"]),
        p(out["cp"]), p(out["P"]), p(out["n"]), p(out["kept"]), p(out["deleted"]), p(out["should_marg"]),
        p(out["slam_to_plane"]), p(out["new_p"]), p(out["new_id"]))
    if rc != 0:
        raise RuntimeError("ovph_run_updater failed with %d" % rc)
    n2 = int(out["n"][0])
    out["n"] = n2
    out["P"] = np.ascontiguousarray(out["P"].reshape(-1)[: n2 * n2].reshape(n2, n2).T)
    out["slam_p"] = out["slam_p"][:n_slam]
    out["cp"] = out["cp"][:n_pl_in]
    out["should_marg"] = out["should_marg"][:n_slam].astype(bool)
    for k in ("kept", "deleted"):
        out[k] = out[k].astype(bool)
    return out


def run_propagate(sc, x, imu, t_state, timestamp, calib_dt, po):
    """Drives ov_plane::Propagator::propagate_and_clone (C++ host mirror; covariance on the device) on the clone window and
    covariance of a synth scene (no planes / SLAM) with IMU state x (dict: q p v bg ba + *_fej) and readings imu [n,7]."""
    L = lib()
    f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)  # noqa: E731
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    N = int(sc.N)
    x16 = f64(np.concatenate([x["q"], x["p"], x["v"], x["bg"], x["ba"]]))
    x16f = f64(np.concatenate([x["q_fej"], x["p_fej"], x["v_fej"], x["bg_fej"], x["ba_fej"]]))
    P = np.asfortranarray(sc.P)
    imu = f64(imu)
    sig = f64([po["sigma_w"], po["sigma_a"], po["sigma_wb"], po["sigma_ab"]])
    cq, cp_ = f64(sc.clone_q), f64(sc.clone_p)
    out = dict(x16=np.zeros(16), x16_fej=np.zeros(16), Phi=np.zeros((15, 15)), Q=np.zeros((15, 15)), last_w=np.zeros(3),
               P=np.zeros((N + 6, N + 6)), new_clone=np.zeros(7))
    L.ovph_run_propagate.restype = C.c_int
    rc = L.ovph_run_propagate(C.c_int(sc.C), p(cq), p(cp_), p(x16), p(x16f), C.c_double(calib_dt), C.c_int(N), p(P),
                              C.c_int(imu.shape[0]), p(imu), C.c_double(t_state), C.c_double(timestamp), p(sig),
                              C.c_double(po["gravity_mag"]), C.c_int(int(po["use_rk4"])), C.c_int(int(po["imu_avg"])),
                              C.c_int(int(po["do_fej"])), p(out["x16"]), p(out["x16_fej"]), p(out["Phi"]), p(out["Q"]),
                              p(out["last_w"]), p(out["P"]), p(out["new_clone"]))
    if rc != 0:
        raise RuntimeError("ovph_run_propagate failed with %d" % rc)
    out["Phi"] = np.ascontiguousarray(out["Phi"].T)   # column-major on the C++ side
    out["Q"] = np.ascontiguousarray(out["Q"].T)
    out["P"] = np.ascontiguousarray(out["P"].T)
    return out


def run_zupt(sc, x, imu, t_state, timestamps, calib_dt, po, noise_mult=10.0, chi2_mult=1.0, max_velocity=0.5, max_disparity=1.0,
             uv0=None, uv1=None):
    """Drives ov_plane::UpdaterZeroVelocity::try_update (C++ host mirror; covariance on the device) at the camera times
    `timestamps` (1 or 2) on the clone window / covariance of a synth scene with IMU state x and readings imu [n,7]; uv0 / uv1
    [n,2]: pixel positions of the tracks the feature database holds at t_state and at the new camera times."""
    L = lib()
    f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)  # noqa: E731
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    N = int(sc.N)
    x16 = f64(np.concatenate([x["q"], x["p"], x["v"], x["bg"], x["ba"]]))
    x16f = f64(np.concatenate([x["q_fej"], x["p_fej"], x["v_fej"], x["bg_fej"], x["ba_fej"]]))
    P = np.asfortranarray(sc.P)
    imu = f64(imu)
    ts = f64(timestamps)
    sig = f64([po["sigma_w"], po["sigma_a"], po["sigma_wb"], po["sigma_ab"]])
    cq, cp_ = f64(sc.clone_q), f64(sc.clone_p)
    uv0 = np.zeros((0, 2), np.float32) if uv0 is None else np.ascontiguousarray(uv0, dtype=np.float32)
    uv1 = np.zeros((0, 2), np.float32) if uv1 is None else np.ascontiguousarray(uv1, dtype=np.float32)
    acc = np.zeros(len(ts), dtype=np.int32)
    chi2 = np.zeros(len(ts))
    out = dict(x16=np.zeros(16), P=np.zeros((N, N)))
    dt_out, t_out, n_meas = C.c_double(0), C.c_double(0), C.c_int(0)
    L.ovph_run_zupt.restype = C.c_int
    rc = L.ovph_run_zupt(C.c_int(sc.C), p(cq), p(cp_), p(x16), p(x16f), C.c_double(calib_dt), C.c_int(N), p(P),
                         C.c_int(imu.shape[0]), p(imu), C.c_double(t_state), C.c_int(len(ts)), p(ts), p(sig),
                         C.c_double(po["gravity_mag"]), C.c_int(int(po["do_fej"])), C.c_double(noise_mult), C.c_double(chi2_mult),
                         C.c_double(max_velocity), C.c_double(max_disparity), C.c_int(uv0.shape[0]), p(uv0), p(uv1), p(acc),
                         p(chi2), p(out["x16"]), C.byref(dt_out), p(out["P"]), C.byref(t_out), C.byref(n_meas))
    if rc != 0:
        raise RuntimeError("ovph_run_zupt failed with %d" % rc)
    out["P"] = np.ascontiguousarray(out["P"].T)
    out.update(accepted=acc.astype(bool), chi2=chi2, calib_dt=dt_out.value, timestamp=t_out.value, meas_at_t1=n_meas.value)
    return out


def run_state_maintenance(sc, should_marg, merge_pairs, active_planes, sigma_plane_merge=0.001, plane_merge_chi2=1.0,
                          plane_merge_deg_max=1.0):
    """StateHelper::marginalize_slam followed by merge_planes_and_marginalize on a make_slam_scene-like state
    (n_slam landmarks + all planes in the state)."""
    L = lib()
    f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)  # noqa: E731
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    N = int(sc.N)
    n_slam, n_pl = int(sc.slam_p.shape[0]), int(sc.cp.shape[0])
    P = np.asfortranarray(sc.P)
    sm = np.ascontiguousarray(should_marg, dtype=np.uint8)
    mp = np.ascontiguousarray(np.asarray(merge_pairs, dtype=np.int32).reshape(-1, 2))
    ap = np.ascontiguousarray(active_planes, dtype=np.int32)
    out = dict(P=np.zeros((N, N)), n=np.zeros(1, dtype=np.int32), plane_id=np.zeros(n_pl + 8, dtype=np.int32),
               plane_cp=np.zeros((n_pl + 8, 3)), slam_id=np.zeros(max(n_slam, 1), dtype=np.int32),
               slam_to_plane=np.zeros(max(n_slam, 1), dtype=np.int32))
    cq, cp_, calq, calp, intr = f64(sc.clone_q), f64(sc.clone_p), f64(sc.calib_q), f64(sc.calib_p), f64(sc.intr)
    slam_p, cp = f64(sc.slam_p if n_slam else np.zeros((1, 3))), f64(sc.cp if n_pl else np.zeros((1, 3)))
    L.ovph_run_state_maintenance.restype = C.c_int
    rc = L.ovph_run_state_maintenance(
        C.c_int(sc.C), p(cq), p(cp_), p(calq), p(calp), p(intr), C.c_int(n_slam), p(slam_p), p(sm), C.c_int(n_pl), p(cp),
        C.c_int(N), p(P), C.c_int(mp.shape[0]), p(mp), C.c_int(len(ap)), p(ap), C.c_double(sigma_plane_merge),
        C.c_double(plane_merge_chi2), C.c_double(plane_merge_deg_max), p(out["P"]), p(out["n"]), p(out["plane_id"]),
        p(out["plane_cp"]), p(out["slam_id"]), p(out["slam_to_plane"]))
    if rc != 0:
        raise RuntimeError("ovph_run_state_maintenance failed with %d" % rc)
    n2 = int(out["n"][0])
    out["n"] = n2
    out["P"] = np.ascontiguousarray(out["P"].reshape(-1)[: n2 * n2].reshape(n2, n2).T)
    out["slam_id"] = out["slam_id"][:n_slam]
    out["slam_to_plane"] = out["slam_to_plane"][:n_slam]
    return out


def run_plane_givens(op, H_f, H_x, H_cp, res):
    """UpdaterPlane / UpdaterHelper ::nullspace_project_inplace (op 0) or ::measurement_compress_inplace (op 1) of the C++
    host mirror on dense matrices; H_cp None selects the UpdaterHelper version."""
    L = lib()
    rows, cols = H_x.shape
    Hx = np.asfortranarray(H_x, dtype=np.float64).copy(order="F")
    Hf = np.asfortranarray(H_f if H_f is not None else np.zeros((rows, 1)), dtype=np.float64).copy(order="F")
    Hc = np.asfortranarray(H_cp if H_cp is not None else np.zeros((rows, 1)), dtype=np.float64).copy(order="F")
    r = np.ascontiguousarray(res, dtype=np.float64).copy()
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    L.ovph_run_plane_givens.restype = C.c_int
    ro = L.ovph_run_plane_givens(C.c_int(op), C.c_int(rows), C.c_int(0 if H_f is None else H_f.shape[1]), p(Hf), C.c_int(cols),
                                 p(Hx), C.c_int(0 if H_cp is None else H_cp.shape[1]), p(Hc), p(r))
    return Hx[:ro].copy(), (Hc[:ro].copy() if H_cp is not None else None), r[:ro].copy()


def run_sequence(init, imu, frame_time, frames, po, sigma_px=1.0, chi2_mult=1.0, trace=False, plane_mode=0, plane_min_feat=20,
                 sigma_c=0.01):
    """Closed loop over several frames through the C++ host mirror (Propagator::propagate_and_clone ->
    UpdaterMSCKF::update -> StateHelper::marginalize_old_clone per frame).

    init: dict(C, N, clone_q, clone_p, clone_q_fej, clone_p_fej, calib_q, calib_p, intr, x (IMU dict), dt, P, t_state)
    frames: list of dict(uv [F,M,2] f32, slot [F,M] int32 (index into the C+1 clones of the window), n_meas [F], p_FinG [F,3])
    A frame may carry `uv_norm` [F,M,2] instead of p_FinG (all frames or none): the updater then triangulates its features.
    trace: also return the IMU value ("traj" [K,16]) and its pose covariance ("posecov" [K,6,6]) after every frame.
    plane_mode 1 / 2 with per-frame `plane` [F] (0 = free point): planar regularities in UpdaterMSCKF / additionally
    UpdaterPlane::init_vio_plane (the planes join the state; the final P is then not returned, "planes_in_state" is).
    """
    L = lib()
    f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)  # noqa: E731
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    Cn, N = int(init["C"]), int(init["N"])
    x = init["x"]
    x16 = f64(np.concatenate([x["q"], x["p"], x["v"], x["bg"], x["ba"]]))
    x16f = f64(np.concatenate([x["q_fej"], x["p_fej"], x["v_fej"], x["bg_fej"], x["ba_fej"]]))
    M = max(int(fr["uv"].shape[1]) for fr in frames)
    offs = np.zeros(len(frames) + 1, dtype=np.int32)
    uv_l, slot_l, nm_l, pf_l, uvn_l = [], [], [], [], []
    with_norm = all("uv_norm" in fr for fr in frames)
    for k, fr in enumerate(frames):
        F = int(fr["uv"].shape[0])
        offs[k + 1] = offs[k] + F
        uvp = np.zeros((F, M, 2), dtype=np.float32)
        uvp[:, : fr["uv"].shape[1]] = fr["uv"]
        sl = -np.ones((F, M), dtype=np.int32)
        sl[:, : fr["slot"].shape[1]] = fr["slot"]
        uv_l.append(uvp)
        slot_l.append(sl)
        nm_l.append(np.asarray(fr["n_meas"], dtype=np.int32))
        pf_l.append(np.asarray(fr["p_FinG"], dtype=np.float64) if "p_FinG" in fr else np.zeros((F, 3)))
        if with_norm:
            un = np.zeros((F, M, 2), dtype=np.float32)
            un[:, : fr["uv_norm"].shape[1]] = fr["uv_norm"]
            uvn_l.append(un)
    uv = np.ascontiguousarray(np.concatenate(uv_l), dtype=np.float32)
    slot = np.ascontiguousarray(np.concatenate(slot_l), dtype=np.int32)
    nm = np.ascontiguousarray(np.concatenate(nm_l), dtype=np.int32)
    pf = f64(np.concatenate(pf_l))
    P = np.asfortranarray(init["P"])
    imu = f64(imu)
    ft = f64(frame_time)
    sig = f64([po["sigma_w"], po["sigma_a"], po["sigma_wb"], po["sigma_ab"]])
    cq, cp_, cqf, cpf = f64(init["clone_q"]), f64(init["clone_p"]), f64(init["clone_q_fej"]), f64(init["clone_p_fej"])
    calq, calp, intr = f64(init["calib_q"]), f64(init["calib_p"]), f64(init["intr"])
    out = dict(clone_q=np.zeros((Cn, 4)), clone_p=np.zeros((Cn, 3)), x16=np.zeros(16), calib_q=np.zeros(4), calib_p=np.zeros(3),
               intr=np.zeros(8), dt=np.zeros(1), P=np.zeros((N, N)), kept=np.zeros(len(frames), dtype=np.int32))
    K = len(frames)
    traj, posecov = np.zeros((K, 16)), np.zeros((K, 36))
    uvn = np.ascontiguousarray(np.concatenate(uvn_l), dtype=np.float32) if with_norm else None
    if trace or with_norm:
        L.ovph_set_sequence_trace(p(traj) if trace else None, p(posecov) if trace else None, p(uvn) if with_norm else None)
    n_in_state = C.c_int(0)
    if plane_mode:
        pl = np.ascontiguousarray(np.concatenate([np.asarray(fr["plane"], dtype=np.int32) for fr in frames]), dtype=np.int32)
        L.ovph_set_sequence_planes(p(pl), C.c_int(int(plane_mode)), C.c_int(int(plane_min_feat)), C.c_double(sigma_c),
                                   C.byref(n_in_state))
    L.ovph_run_sequence.restype = C.c_int
    rc = L.ovph_run_sequence(
        C.c_int(Cn), p(cq), p(cp_), p(cqf), p(cpf), p(calq), p(calp), p(intr), p(x16), p(x16f), C.c_double(init["dt"]),
        C.c_int(N), p(P), C.c_int(imu.shape[0]), p(imu), C.c_double(init["t_state"]), p(sig), C.c_double(po["gravity_mag"]),
        C.c_int(int(po["use_rk4"])), C.c_int(int(po["do_fej"])), C.c_int(len(frames)), p(ft), p(offs), C.c_int(M), p(uv),
        p(slot), p(nm), p(pf), C.c_double(sigma_px), C.c_double(chi2_mult), p(out["clone_q"]), p(out["clone_p"]), p(out["x16"]),
        p(out["calib_q"]), p(out["calib_p"]), p(out["intr"]), p(out["dt"]), p(out["P"]), p(out["kept"]))
    if rc != 0:
        raise RuntimeError("ovph_run_sequence failed with %d" % rc)
    out["P"] = np.ascontiguousarray(out["P"].T)
    out["dt"] = float(out["dt"][0])
    out["planes_in_state"] = n_in_state.value
    if trace:
        out["traj"] = traj
        out["posecov"] = posecov.reshape(K, 6, 6)   # symmetric: the column-major layout does not matter
    return out


def feature_jacobian_rep(sc, f, rep, anchor_ci):
    """ov_plane::UpdaterHelper::get_feature_jacobian_full (C++ host mirror) for feature f of a synth scene held in landmark
    representation rep (0..5) anchored in clone slot anchor_ci.  Returns (H_f, H_x, res, order)."""
    L = lib()
    f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)  # noqa: E731
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    m = int(sc.n_meas[f])
    N = int(sc.N)
    P = np.asfortranarray(sc.P)
    uv = np.ascontiguousarray(sc.uv[f, :m], dtype=np.float32)
    cidx = np.ascontiguousarray(sc.clone_idx[f, :m], dtype=np.int32)
    pf = f64(sc.p_FinG[f])
    cq, cp_, cqf, cpf_ = f64(sc.clone_q), f64(sc.clone_p), f64(sc.clone_q_fej), f64(sc.clone_p_fej)
    calq, calp, intr = f64(sc.calib_q), f64(sc.calib_p), f64(sc.intr)
    maxr, maxc = 2 * m, 6 * m + 20
    H_f, H_x, res = np.zeros(maxr * 3), np.zeros(maxr * maxc), np.zeros(maxr)
    rows, cols, hfc, no = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    oid, osz = np.zeros(m + 4, dtype=np.int32), np.zeros(m + 4, dtype=np.int32)
    L.ovph_feature_jacobian_rep.restype = C.c_int
    rc = L.ovph_feature_jacobian_rep(C.c_int(sc.C), p(cq), p(cp_), p(cqf), p(cpf_), p(calq), p(calp), p(intr), C.c_int(N), p(P),
                                     C.c_int(m), p(uv), p(cidx), p(pf), C.c_int(rep), C.c_int(anchor_ci),
                                     C.c_double(sc.opts["sigma_px"]), C.c_int(int(sc.opts["do_fej"])), p(H_f), p(H_x), p(res),
                                     C.byref(rows), C.byref(cols), C.byref(hfc), p(oid), p(osz), C.byref(no))
    if rc != 0:
        raise RuntimeError("ovph_feature_jacobian_rep failed with %d" % rc)
    r, c, h, n = rows.value, cols.value, hfc.value, no.value
    return (H_f[: r * h].reshape(h, r).T.copy(), H_x[: r * c].reshape(c, r).T.copy(), res[:r].copy(),
            [(int(oid[i]), int(osz[i])) for i in range(n)])


def run_change_anchors(sc, rep, p_FinA, p_FinA_fej):
    """Drives ov_plane::UpdaterSLAM::change_anchors (C++ host mirror) on a make_scene(n_slam >= 1) state whose first landmark
    is held in the anchored representation rep and anchored in the oldest clone.  Returns dict(value, fej, anchor_ci, P)."""
    L = lib()
    f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)  # noqa: E731
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    N = int(sc.N)
    P = np.asfortranarray(sc.P)
    cq, cp_, cqf, cpf_ = f64(sc.clone_q), f64(sc.clone_p), f64(sc.clone_q_fej), f64(sc.clone_p_fej)
    calq, calp, intr, slam = f64(sc.calib_q), f64(sc.calib_p), f64(sc.intr), f64(sc.slam_p)
    pa, pf = f64(p_FinA), f64(p_FinA_fej)
    out = dict(value=np.zeros(3), fej=np.zeros(3), P=np.zeros((N, N)))
    anchor = C.c_int(-1)
    L.ovph_run_change_anchors.restype = C.c_int
    rc = L.ovph_run_change_anchors(C.c_int(sc.C), p(cq), p(cp_), p(cqf), p(cpf_), p(calq), p(calp), p(intr), C.c_int(len(slam)),
                                   p(slam), C.c_int(N), p(P), C.c_int(rep), p(pa), p(pf), C.c_int(int(sc.opts["do_fej"])),
                                   p(out["value"]), p(out["fej"]), C.byref(anchor), p(out["P"]))
    if rc != 0:
        raise RuntimeError("ovph_run_change_anchors failed with %d" % rc)
    out["anchor_ci"] = anchor.value
    out["P"] = np.ascontiguousarray(out["P"].T)
    return out


def load_trajectory(path, cap=200000):
    """Trajectory file (`t tx ty tz qx qy qz qw`, '#' comments) through the C++ reader of csrc/host/ov_plane_io.cpp
    (sim/Simulator.cpp load_data; data/udel_arl_short.txt is such a file).  Returns [n, 8]."""
    out = np.zeros((cap, 8))
    L = lib()
    L.ovph_load_trajectory.restype = C.c_int
    n = L.ovph_load_trajectory(str(path).encode(), out.ctypes.data_as(C.c_void_p), C.c_int(cap))
    if n < 0:
        raise FileNotFoundError(path)
    return out[:min(n, cap)].copy()


class Session:
    """A filter session of the C++ host mirror (csrc/host/ov_plane_session.cpp: propagate -> marginalise lost landmarks ->
    plane init -> MSCKF update -> SLAM update -> SLAM delayed init -> anchor change -> marginalise the oldest clone), the
    covariance resident on the device between frames.  init: the dict of closed_loop.initial_state."""

    def __init__(self, init, po, sigma_px=1.0, chi2_mult=1.0, chi2_mult_slam=1.0, plane_mode=0, plane_min_feat=20, sigma_c=0.01,
                 max_slam=0, feat_rep_slam=0, cam_dt=0.1):
        L = lib()
        f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)  # noqa: E731
        p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        x = init["x"]
        x16 = f64(np.concatenate([x["q"], x["p"], x["v"], x["bg"], x["ba"]]))
        oi = np.array([int(po["use_rk4"]), int(po["do_fej"]), plane_mode, plane_min_feat, max_slam, feat_rep_slam], dtype=np.int32)
        od = f64([po["sigma_w"], po["sigma_a"], po["sigma_wb"], po["sigma_ab"], po["gravity_mag"], sigma_px, chi2_mult,
                  chi2_mult_slam, sigma_c, cam_dt])
        P = np.asfortranarray(init["P"])
        L.ovph_session_open.restype = C.c_void_p
        self._L = L
        self.C = int(init["C"])
        self.max_slam = max_slam
        self._h = L.ovph_session_open(C.c_int(self.C), p(f64(init["clone_q"])), p(f64(init["clone_p"])), p(f64(init["calib_q"])),
                                      p(f64(init["calib_p"])), p(f64(init["intr"])), p(x16), C.c_double(init["dt"]),
                                      C.c_int(int(init["N"])), p(P), C.c_double(init["t_state"]), p(oi), p(od))
        if not self._h:
            raise RuntimeError("ovph_session_open failed")

    def feed_imu(self, imu):
        imu = np.ascontiguousarray(imu, dtype=np.float64).reshape(-1, 7)
        self._L.ovph_session_feed_imu(C.c_void_p(self._h), C.c_int(imu.shape[0]), imu.ctypes.data_as(C.c_void_p))

    def enable_zupt(self, po, max_velocity=0.1, noise_multiplier=10.0, max_disparity=0.5, chi2_mult=1.0):
        """UpdaterZeroVelocity in front of every frame (VioManagerOptions::try_zupt and its zupt_* values); call before
        feed_imu so the detector sees the readings too."""
        f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)  # noqa: E731
        z = f64([max_velocity, noise_multiplier, max_disparity, chi2_mult])
        sg = f64([po["sigma_w"], po["sigma_a"], po["sigma_wb"], po["sigma_ab"]])
        self._L.ovph_session_enable_zupt(C.c_void_p(self._h), z.ctypes.data_as(C.c_void_p), sg.ctypes.data_as(C.c_void_p),
                                         C.c_double(po["gravity_mag"]))

    def feed_tracks(self, frame_time, fid, uv):
        """raw pixel tracks of a frame for the detector's disparity test (ext FeatureDatabase)"""
        fid = np.ascontiguousarray(fid, dtype=np.int64)
        uv = np.ascontiguousarray(uv, dtype=np.float32).reshape(-1, 2)
        self._L.ovph_session_feed_tracks(C.c_void_p(self._h), C.c_double(frame_time), C.c_int(len(fid)),
                                         fid.ctypes.data_as(C.c_void_p), uv.ctypes.data_as(C.c_void_p))

    def try_zupt(self, frame_time):
        """None when the frame has to be processed normally, else dict(x16, posecov, chi2): the zero-velocity update was
        applied and the frame is not cloned."""
        x16, pc, chi2 = np.zeros(16), np.zeros(36), C.c_double(0)
        self._L.ovph_session_try_zupt.restype = C.c_int
        did = self._L.ovph_session_try_zupt(C.c_void_p(self._h), C.c_double(frame_time), x16.ctypes.data_as(C.c_void_p),
                                            pc.ctypes.data_as(C.c_void_p), C.byref(chi2))
        return dict(x16=x16, posecov=pc.reshape(6, 6), chi2=chi2.value) if did else None

    def open_files(self, est=None, std=None, gt=None, timing=None):
        """Output files in the reference's formats (state estimate / standard deviations / groundtruth, timing CSV)."""
        enc = lambda s: (s or "").encode()  # noqa: E731
        if self._L.ovph_session_open_files(C.c_void_p(self._h), enc(est), enc(std), enc(gt), enc(timing)) != 0:
            raise RuntimeError("ovph_session_open_files failed")

    def step(self, frame_time, uv, uv_norm, slot, n_meas, fid, kind, plane=None, truth=None, active_planes=None, merged_planes=None):
        """One camera frame (arrays as in ovph_session_step2).  Returns dict(counts, x16, posecov [6,6], slam_ids).
        truth [33]: simulator state, time offset, intrinsics and extrinsics for the groundtruth file (open_files).
        active_planes: ids of the planes the tracker currently sees (over all live tracks); planes in the state that are not among
        them are marginalised, merged_planes [(surviving id, old id), ...] are fused first (core/VioManager.cpp:513-534 ->
        StateHelper::merge_planes_and_marginalize).  None = no plane leaves the state."""
        p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        F = int(len(n_meas))
        M = int(uv.shape[1]) if F else 1
        uv = np.ascontiguousarray(uv, dtype=np.float32).reshape(F, M, 2)
        uvn = np.ascontiguousarray(uv_norm, dtype=np.float32).reshape(F, M, 2)
        slot = np.ascontiguousarray(slot, dtype=np.int32).reshape(F, M)
        nm = np.ascontiguousarray(n_meas, dtype=np.int32)
        gf = np.ascontiguousarray(fid, dtype=np.int64)
        kd = np.ascontiguousarray(kind, dtype=np.int32)
        pl = np.ascontiguousarray(plane if plane is not None else np.zeros(F), dtype=np.int32)
        counts = np.zeros(6, dtype=np.int32)
        x16, pc = np.zeros(16), np.zeros(36)
        cap = max(self.max_slam, 1) + 8
        ids = -np.ones(cap, dtype=np.int64)
        act = None if active_planes is None else np.ascontiguousarray(sorted(set(int(a) for a in active_planes)), dtype=np.int64)
        mrg = np.ascontiguousarray(merged_planes if merged_planes is not None else np.zeros((0, 2)), dtype=np.int64).reshape(-1, 2)
        self._L.ovph_session_step2.restype = C.c_int
        rc = self._L.ovph_session_step2(C.c_void_p(self._h), C.c_double(frame_time), C.c_int(F), C.c_int(M), p(uv), p(uvn), p(slot),
                                        p(nm), p(gf), p(kd), p(pl), p(counts), p(x16), p(pc), C.c_int(cap), p(ids),
                                        p(np.ascontiguousarray(truth, dtype=np.float64)) if truth is not None else None,
                                        C.c_int(-1 if act is None else len(act)), p(act) if (act is not None and len(act)) else None,
                                        C.c_int(len(mrg)), p(mrg) if len(mrg) else None)
        if rc != 0:
            raise RuntimeError("ovph_session_step failed with %d" % rc)
        return dict(counts=counts, x16=x16, posecov=pc.reshape(6, 6), slam_ids=[int(i) for i in ids[:counts[4]]])

    def close(self):
        if self._h:
            self._L.ovph_session_close(C.c_void_p(self._h))
            self._h = None
