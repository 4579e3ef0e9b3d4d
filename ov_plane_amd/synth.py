"""Deterministic synthetic clone/feature/plane batches for the MSCKF(+plane) update path.

The generator imitates the reference simulator's input distribution:
  * camera/IMU calibration of config/sim/kalibr_imucam_chain.yaml cam0 (radtan, 752x480)
  * clones at 10 Hz (config/sim/estimator_config.yaml: sim_freq_cam), features at 2..5 m
    (sim_min/max_feature_gen_dist), sigma_px = 1 (up_msckf_sigma_px), uv stored as f32
    (ov_plane/src/update/UpdaterHelper.h:68)
  * state order of ov_plane/src/state/State.cpp:33-82: IMU(15) | dt(1) | calib pose(6) | intrinsics(8) | clones(6 each)
    [| SLAM landmarks (3 each) | planes (3 each)]
It is *input generation only*: nothing here is on the measured path, and nothing here comes from /root/reference
at run time (the numbers above are constants of the reference's public sim config).
"""
from __future__ import annotations

import numpy as np

# ----------------------------------------------------------------------------------------------
# calibration constants (config/sim/kalibr_imucam_chain.yaml cam0)
# ----------------------------------------------------------------------------------------------
T_IMU_CAM = np.array(
    [
        [0.0148655429818, -0.999880929698, 0.00414029679422, -0.0216401454975],
        [0.999557249008, 0.0149672133247, 0.025715529948, -0.064676986768],
        [-0.0257744366974, 0.00375618835797, 0.999660727178, 0.00981073058949],
        [0.0, 0.0, 0.0, 1.0],
    ]
)
INTRINSICS = np.array([458.654, 457.296, 367.215, 248.375, -0.28340811, 0.07395907, 0.00019359, 1.76187114e-05])
IMG_W, IMG_H = 752.0, 480.0


# ----------------------------------------------------------------------------------------------
# JPL quaternion helpers (ext ov_core quat_ops.h semantics)
# ----------------------------------------------------------------------------------------------
def skew(w):
    return np.array([[0.0, -w[2], w[1]], [w[2], 0.0, -w[0]], [-w[1], w[0], 0.0]])


def quat_2_rot(q):
    qv = np.asarray(q[:3], dtype=np.float64)
    q4 = float(q[3])
    return (2.0 * q4 * q4 - 1.0) * np.eye(3) - 2.0 * q4 * skew(qv) + 2.0 * np.outer(qv, qv)


def rot_2_quat(R):
    """JPL rot_2_quat (4-branch), q = [x y z w] with R = quat_2_rot(q)."""
    T = np.trace(R)
    q = np.zeros(4)
    if R[0, 0] >= T and R[0, 0] >= R[1, 1] and R[0, 0] >= R[2, 2]:
        q[0] = np.sqrt((1 + 2 * R[0, 0] - T) / 4)
        q[1] = (1 / (4 * q[0])) * (R[0, 1] + R[1, 0])
        q[2] = (1 / (4 * q[0])) * (R[0, 2] + R[2, 0])
        q[3] = (1 / (4 * q[0])) * (R[1, 2] - R[2, 1])
    elif R[1, 1] >= T and R[1, 1] >= R[0, 0] and R[1, 1] >= R[2, 2]:
        q[1] = np.sqrt((1 + 2 * R[1, 1] - T) / 4)
        q[0] = (1 / (4 * q[1])) * (R[0, 1] + R[1, 0])
        q[2] = (1 / (4 * q[1])) * (R[1, 2] + R[2, 1])
        q[3] = (1 / (4 * q[1])) * (R[2, 0] - R[0, 2])
    elif R[2, 2] >= T and R[2, 2] >= R[0, 0] and R[2, 2] >= R[1, 1]:
        q[2] = np.sqrt((1 + 2 * R[2, 2] - T) / 4)
        q[0] = (1 / (4 * q[2])) * (R[0, 2] + R[2, 0])
        q[1] = (1 / (4 * q[2])) * (R[1, 2] + R[2, 1])
        q[3] = (1 / (4 * q[2])) * (R[0, 1] - R[1, 0])
    else:
        q[3] = np.sqrt((1 + T) / 4)
        q[0] = (1 / (4 * q[3])) * (R[1, 2] - R[2, 1])
        q[1] = (1 / (4 * q[3])) * (R[2, 0] - R[0, 2])
        q[2] = (1 / (4 * q[3])) * (R[0, 1] - R[1, 0])
    if q[3] < 0:
        q = -q
    return q / np.linalg.norm(q)


def quat_multiply(q, p):
    """JPL q (x) p, then sign/unit normalised."""
    qv, q4 = q[:3], q[3]
    Qm = np.zeros((4, 4))
    Qm[:3, :3] = q4 * np.eye(3) - skew(qv)
    Qm[:3, 3] = qv
    Qm[3, :3] = -qv
    Qm[3, 3] = q4
    out = Qm @ p
    if out[3] < 0:
        out = -out
    return out / np.linalg.norm(out)


def quat_boxplus(q, dth):
    """JPLQuat::update: dq = quatnorm([0.5*dth, 1]); q <- dq (x) q."""
    dq = np.array([0.5 * dth[0], 0.5 * dth[1], 0.5 * dth[2], 1.0])
    dq = dq / np.linalg.norm(dq)
    return quat_multiply(dq, q)


def rotz(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])


def rotx(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[1.0, 0.0, 0.0], [0.0, c, -s], [0.0, s, c]])


def roty(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]])


# ----------------------------------------------------------------------------------------------
# camera model (radtan) - vectorised; used only to synthesise measurements / triangulate
# ----------------------------------------------------------------------------------------------
def radtan_distort(xn, yn, intr):
    fx, fy, cx, cy, k1, k2, p1, p2 = intr
    r2 = xn * xn + yn * yn
    g = 1.0 + k1 * r2 + k2 * r2 * r2
    x1 = xn * g + 2.0 * p1 * xn * yn + p2 * (r2 + 2.0 * xn * xn)
    y1 = yn * g + p1 * (r2 + 2.0 * yn * yn) + 2.0 * p2 * xn * yn
    return fx * x1 + cx, fy * y1 + cy


FISHEYE_INTRINSICS = np.array([380.0, 379.2, 367.215, 248.375, 0.0034823894, 0.0007150348, -0.0020532361, 0.00020293673])


def equi_distort(xn, yn, intr):
    """ext ov_core CamEqui (fisheye): theta_d = theta + k1 theta^3 + k2 theta^5 + k3 theta^7 + k4 theta^9."""
    fx, fy, cx, cy, k1, k2, k3, k4 = intr
    r = np.sqrt(xn * xn + yn * yn)
    th = np.arctan(r)
    th_d = th + k1 * th**3 + k2 * th**5 + k3 * th**7 + k4 * th**9
    cdist = np.where(r > 1e-8, th_d / np.maximum(r, 1e-300), 1.0)
    return fx * xn * cdist + cx, fy * yn * cdist + cy


def equi_undistort(u, v, intr, iters=30):
    """Inverse of the equidistant model (Newton on theta, as cv::fisheye::undistortPoints)."""
    fx, fy, cx, cy, k1, k2, k3, k4 = [float(a) for a in intr]
    x0 = (np.asarray(u, dtype=np.float64) - cx) / fx
    y0 = (np.asarray(v, dtype=np.float64) - cy) / fy
    th_d = np.sqrt(x0 * x0 + y0 * y0)
    th = th_d.copy()
    for _ in range(iters):
        t2 = th * th
        f = th * (1 + k1 * t2 + k2 * t2**2 + k3 * t2**3 + k4 * t2**4) - th_d
        df = 1 + 3 * k1 * t2 + 5 * k2 * t2**2 + 7 * k3 * t2**3 + 9 * k4 * t2**4
        th = th - f / df
    scale = np.where(th_d > 1e-12, np.tan(th) / np.maximum(th_d, 1e-300), 1.0)
    return x0 * scale, y0 * scale


def project_all(p_f, R_GtoI, p_IinG, R_ItoC, p_IinC, intr, fisheye=False):
    """p_f [F,3]; R_GtoI [C,3,3]; p_IinG [C,3]  ->  uv [F,C,2], z [F,C]."""
    d = p_f[:, None, :] - p_IinG[None, :, :]  # F,C,3
    p_I = np.einsum("cij,fcj->fci", R_GtoI, d)
    p_C = np.einsum("ij,fcj->fci", R_ItoC, p_I) + p_IinC[None, None, :]
    z = p_C[..., 2]
    xn = p_C[..., 0] / z
    yn = p_C[..., 1] / z
    u, v = (equi_distort if fisheye else radtan_distort)(xn, yn, intr)
    return np.stack([u, v], axis=-1), z


def radtan_undistort(u, v, intr, iters=30):
    """Inverse of the radtan model by fixed-point iteration (what the tracker's undistort_cv does before a feature's
    uvs_norm are stored, ext ov_core CamRadtan / cv::undistortPoints).  u, v arrays of raw pixels -> normalised coordinates."""
    fx, fy, cx, cy, k1, k2, p1, p2 = [float(a) for a in intr]
    x0 = (np.asarray(u, dtype=np.float64) - cx) / fx
    y0 = (np.asarray(v, dtype=np.float64) - cy) / fy
    x, y = x0.copy(), y0.copy()
    for _ in range(iters):
        r2 = x * x + y * y
        g = 1 + k1 * r2 + k2 * r2 * r2
        dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
        dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
        x = (x0 - dx) / g
        y = (y0 - dy) / g
    return x, y


# ----------------------------------------------------------------------------------------------
# scene container
# ----------------------------------------------------------------------------------------------
class Scene(dict):
    """Plain dict with attribute access. Keys documented in make_scene()."""

    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def state_layout(C, n_slam=0, n_planes_in_state=0):
    ids = dict(imu=0, dt=15, calib=16, intr=22)
    ids["clones"] = np.array([30 + 6 * i for i in range(C)], dtype=np.int32)
    base = 30 + 6 * C
    ids["slam"] = np.array([base + 3 * i for i in range(n_slam)], dtype=np.int32)
    base += 3 * n_slam
    ids["planes"] = np.array([base + 3 * i for i in range(n_planes_in_state)], dtype=np.int32)
    base += 3 * n_planes_in_state
    ids["N"] = base
    return ids


def _trajectory(C, rng):
    """C poses at 10 Hz: lateral motion ~0.8 m/s with the camera tracking a point ~3.5 m ahead."""
    R_CtoI = T_IMU_CAM[:3, :3]
    # camera axes in G at zero yaw: z_C -> +x_G, x_C -> -y_G, y_C -> -z_G
    R_CtoG0 = np.array([[0.0, 0.0, 1.0], [-1.0, 0.0, 0.0], [0.0, -1.0, 0.0]])
    Rs, ps = [], []
    ph = rng.uniform(0, 2 * np.pi, size=4)
    for i in range(C):
        t = 0.1 * i
        p = np.array([0.25 * np.sin(0.9 * t + ph[0]), 0.8 * t, 0.12 * np.sin(1.3 * t + ph[1])])
        yaw = np.arctan2(1.2 - p[1], 3.5 - p[0])
        roll = 0.04 * np.sin(1.1 * t + ph[2])
        pitch = 0.03 * np.sin(0.7 * t + ph[3])
        R_CtoG = rotz(yaw) @ roty(pitch) @ rotx(roll) @ R_CtoG0
        R_GtoI = R_CtoI @ R_CtoG.T
        Rs.append(R_GtoI)
        ps.append(p)
    return np.array(Rs), np.array(ps)


def _cov(C, ids, rng, n_extra=0):
    """SPD covariance with strong clone cross-correlation (common gauge error + random walk)."""
    N = ids["N"]
    nz = 6 + 6 * C + 64 + N
    B = np.zeros((N, nz))
    # common (gauge-like) error shared by every pose
    s_th_c, s_p_c = 1.0e-2, 3.0e-2
    s_th_w, s_p_w = 2.0e-3, 5.0e-3
    col = 0
    pose_ids = list(ids["clones"]) + [ids["imu"]]  # imu pose ~ newest clone (+ its own walk step)
    for pid in pose_ids:
        B[pid : pid + 3, 0:3] = s_th_c * np.eye(3)
        B[pid + 3 : pid + 6, 3:6] = s_p_c * np.eye(3)
    col = 6
    for i in range(C):  # random-walk increment i affects clones i..C-1 and the imu
        for j in range(i, C):
            cid = ids["clones"][j]
            B[cid : cid + 3, col : col + 3] = s_th_w * np.eye(3)
            B[cid + 3 : cid + 6, col + 3 : col + 6] = s_p_w * np.eye(3)
        B[0:3, col : col + 3] = s_th_w * np.eye(3)
        B[3:6, col + 3 : col + 6] = s_p_w * np.eye(3)
        col += 6
    # generic dense mixing so every block of P is populated
    scale = np.full(N, 1.0e-3)
    scale[6:9] = 2.0e-2  # vel
    scale[9:12] = 1.0e-3  # bg
    scale[12:15] = 1.0e-2  # ba
    scale[ids["dt"]] = 2.0e-3
    scale[ids["calib"] : ids["calib"] + 3] = 3.0e-3
    scale[ids["calib"] + 3 : ids["calib"] + 6] = 5.0e-3
    scale[ids["intr"] : ids["intr"] + 4] = 0.5
    scale[ids["intr"] + 4 : ids["intr"] + 8] = 2.0e-3
    for k in range(len(ids["slam"])):
        scale[ids["slam"][k] : ids["slam"][k] + 3] = 3.0e-2
    for k in range(len(ids["planes"])):
        scale[ids["planes"][k] : ids["planes"][k] + 3] = 1.0e-2
    B[:, col : col + 64] = 0.35 * scale[:, None] * rng.standard_normal((N, 64)) / 8.0
    col += 64
    B[:, col : col + N] = np.diag(scale)
    # P = B B^T summed column by column with elementwise operations only: a BLAS product picks its blocking (hence its summation
    # order, hence the last bits of P) by CPU model, and the plane-level chi2 of the reference is decided by exactly such bits
    # (tests/golden/plane_gate_ensemble.npz must mean the same frame on the machine that made it and on the GPU box)
    P = np.zeros((N, N))
    for k in range(B.shape[1]):
        bk = B[:, k]
        nzk = np.nonzero(bk)[0]
        if len(nzk):
            P[np.ix_(nzk, nzk)] += np.multiply.outer(bk[nzk], bk[nzk])
    P = 0.5 * (P + P.T)
    return P


def _triangulate_gn(p0, uv, mask, R_GtoI, p_IinG, R_ItoC, p_IinC, intr, iters=6, fisheye=False):
    """Vectorised Gauss-Newton on reprojection error with the *estimated* poses (stand-in for the
    reference's upstream FeatureInitializer; SURVEY.md §8f rank 1 - input generation only here)."""
    p = p0.copy()
    eps = 1e-6
    for _ in range(iters):
        uv0, _ = project_all(p, R_GtoI, p_IinG, R_ItoC, p_IinC, intr, fisheye)
        r = (uv - uv0) * mask[..., None]  # F,C,2
        J = np.zeros(uv0.shape + (3,))
        for a in range(3):
            dp = np.zeros(3)
            dp[a] = eps
            uva, _ = project_all(p + dp, R_GtoI, p_IinG, R_ItoC, p_IinC, intr, fisheye)
            J[..., a] = (uva - uv0) / eps
        J = J * mask[..., None, None]
        F = p.shape[0]
        Jf = J.reshape(F, -1, 3)
        rf = r.reshape(F, -1)
        A = np.einsum("fka,fkb->fab", Jf, Jf) + 1e-9 * np.eye(3)[None]
        b = np.einsum("fka,fk->fa", Jf, rf)
        p = p + np.linalg.solve(A, b[..., None])[..., 0]
    return p


def make_scene(
    C=11,
    F=200,
    seed=0,
    ragged=False,
    n_planes=0,
    feats_per_plane=50,
    planes_in_state_frac=0.5,
    n_slam=0,
    chi2_mult=1.0,
    sigma_px=1.0,
    sigma_c=0.05,
    do_fej=True,
    calib=True,
    min_meas=5,
    feat_seed=None,
    px_noise=None,
    err_scale=0.85,
    fisheye=False,
):
    """Build one synthetic update-step input.

    Returns Scene with (all float64 unless noted):
      C, F, N, ids (state_layout)
      clone_q/clone_p [C,4]/[C,3] current estimates, clone_q_fej/clone_p_fej first estimates
      imu_q,imu_p,imu_v,imu_bg,imu_ba ; calib_q [4] (R_ItoC), calib_p [3] (p_IinC) ; intr [8] ; dt
      P [N,N]
      uv [F,M,2] float32, clone_idx [F,M] int32 (-1 = padding), n_meas [F] int32, M = C
      p_FinG [F,3] linearisation points (GLOBAL_3D)
      plane_id [F] int32 (0 = free point), planes: cp [n_planes,3] estimates, cp_fej, in_state [n_planes] bool,
             plane_state_id [n_planes] (column offset or -1)
      opts: sigma_px, sigma_c, chi2_mult, do_fej, do_calib_pose, do_calib_intr
      truth: dict of ground-truth quantities (for diagnostics only)
    """
    rng = np.random.default_rng(seed)
    # features / measurement noise may use their own stream so that several ranks can share one filter state
    frng = np.random.default_rng(1000003 + int(seed if feat_seed is None else feat_seed))
    n_in_state = int(round(n_planes * planes_in_state_frac))
    ids = state_layout(C, n_slam=n_slam, n_planes_in_state=n_in_state)
    N = ids["N"]

    R_true, p_true = _trajectory(C, rng)
    R_CtoI = T_IMU_CAM[:3, :3]
    p_CinI = T_IMU_CAM[:3, 3]
    R_ItoC_true = R_CtoI.T
    p_IinC_true = -R_ItoC_true @ p_CinI
    intr_true = (FISHEYE_INTRINSICS if fisheye else INTRINSICS).copy()

    # ---- planes: faces of a box in front of the trajectory (cf. Simulator::generate_planes) ------------
    planes_n, planes_d = [], []
    for k in range(n_planes):
        # alternate: back wall (normal ~ -x), floor (normal ~ +z), ceiling (normal ~ -z), side walls
        kind = k % 4
        jit = 0.08 * rng.standard_normal(3)
        if kind == 0:
            n = np.array([1.0, 0.0, 0.0]) + jit
            d = 4.2 + 0.9 * (k // 4) / max(1, n_planes // 4) + 0.1 * rng.uniform()
        elif kind == 1:
            n = np.array([0.0, 0.0, 1.0]) + jit
            d = -(1.2 + 0.5 * rng.uniform())
        elif kind == 2:
            n = np.array([0.0, 0.0, 1.0]) + jit
            d = 1.3 + 0.5 * rng.uniform()
        else:
            n = np.array([0.25, 1.0, 0.0]) + jit
            d = 3.6 + 0.8 * rng.uniform()
        n = n / np.linalg.norm(n)
        if d < 0:
            n, d = -n, -d
        planes_n.append(n)
        planes_d.append(d)
    planes_n = np.array(planes_n).reshape(-1, 3)
    planes_d = np.array(planes_d)
    cp_true = planes_n * planes_d[:, None]

    # ---- features: rejection-sample points visible in every clone they are assigned to -----------------
    def visible(pf, lo, hi):
        uvp, z = project_all(pf, R_true[lo:hi], p_true[lo:hi], R_ItoC_true, p_IinC_true, intr_true, fisheye)
        ok = (z > 0.5) & (uvp[..., 0] > 15) & (uvp[..., 0] < IMG_W - 15) & (uvp[..., 1] > 15) & (uvp[..., 1] < IMG_H - 15)
        return ok.all(axis=1)

    n_planar = min(F, n_planes * feats_per_plane)
    n_free = F - n_planar
    if ragged:
        n_meas = frng.integers(min(min_meas, C), C + 1, size=F).astype(np.int32)
        start = np.array([frng.integers(0, C - m + 1) for m in n_meas], dtype=np.int32)
    else:
        n_meas = np.full(F, C, dtype=np.int32)
        start = np.zeros(F, dtype=np.int32)

    p_f = np.zeros((F, 3))
    plane_id = np.zeros(F, dtype=np.int32)
    mid = p_true[C // 2]
    for f in range(F):
        lo, hi = int(start[f]), int(start[f] + n_meas[f])
        for _try in range(2000):
            if f < n_free:
                depth = frng.uniform(2.0, 5.0)
                cand = mid + np.array([depth, frng.uniform(-2.5, 2.5), frng.uniform(-1.4, 1.4)])
            else:
                k = (f - n_free) % n_planes
                # random point on plane k near the viewing volume
                base = mid + np.array([frng.uniform(1.8, 5.2), frng.uniform(-2.5, 2.5), frng.uniform(-1.4, 1.4)])
                cand = base - (planes_n[k] @ base - planes_d[k]) * planes_n[k]
            if visible(cand[None], lo, hi)[0] and (cand[0] - mid[0]) > 1.5:
                break
        else:
            raise RuntimeError("could not place feature %d" % f)
        p_f[f] = cand
        if f >= n_free:
            plane_id[f] = 1 + (f - n_free) % n_planes

    clone_idx = -np.ones((F, C), dtype=np.int32)
    mask = np.zeros((F, C))
    for f in range(F):
        m = int(n_meas[f])
        clone_idx[f, :m] = np.arange(start[f], start[f] + m)
        mask[f, start[f] : start[f] + m] = 1.0

    # ---- covariance and a consistent estimation error -------------------------------------------------
    P = _cov(C, ids, rng)
    Lc = np.linalg.cholesky(P)
    # the filter is made slightly conservative (true error = 0.85 sigma) so that ~95 % of the features pass
    # the 0.95 chi-square gate at chi2_mult = 1 despite second-order effects
    err = err_scale * (Lc @ rng.standard_normal(N))

    clone_q = np.zeros((C, 4))
    clone_p = np.zeros((C, 3))
    for i in range(C):
        cid = ids["clones"][i]
        # estimate = truth [-] err  (so that truth = estimate [+] err)
        clone_q[i] = quat_boxplus(rot_2_quat(R_true[i]), -err[cid : cid + 3])
        clone_p[i] = p_true[i] - err[cid + 3 : cid + 6]
    calib_q = calib_p = intr = None
    if calib:
        calib_q = quat_boxplus(rot_2_quat(R_ItoC_true), -err[ids["calib"] : ids["calib"] + 3])
        calib_p = p_IinC_true - err[ids["calib"] + 3 : ids["calib"] + 6]
        intr = intr_true - err[ids["intr"] : ids["intr"] + 8]
    else:
        calib_q = rot_2_quat(R_ItoC_true)
        calib_p = p_IinC_true.copy()
        intr = intr_true.copy()

    # FEJ: value + small perturbation on a random half of the clones (UpdaterHelper.cpp:376-385 branch)
    clone_q_fej = clone_q.copy()
    clone_p_fej = clone_p.copy()
    half = rng.permutation(C)[: C // 2]
    for i in half:
        clone_q_fej[i] = quat_boxplus(clone_q[i], 1e-3 * rng.standard_normal(3))
        clone_p_fej[i] = clone_p[i] + 1e-3 * rng.standard_normal(3)

    # ---- measurements (truth + N(0, sigma_px)), stored as f32 ------------------------------------------
    uv_true, _ = project_all(p_f, R_true, p_true, R_ItoC_true, p_IinC_true, intr_true, fisheye)
    # px_noise: the tracker's actual noise when it differs from the sigma the filter assumes
    uv_noisy = uv_true + (sigma_px if px_noise is None else px_noise) * frng.standard_normal(uv_true.shape)
    uv = np.zeros((F, C, 2), dtype=np.float32)
    for f in range(F):
        m = int(n_meas[f])
        uv[f, :m] = uv_noisy[f, start[f] : start[f] + m].astype(np.float32)

    # normalised measurements as the tracker stores them (undistorted with the current intrinsics estimate, f32)
    uv_norm = np.zeros((F, C, 2), dtype=np.float32)
    xn, yn = (equi_undistort if fisheye else radtan_undistort)(uv[..., 0].astype(np.float64), uv[..., 1].astype(np.float64), intr)
    uv_norm[..., 0] = xn.astype(np.float32)
    uv_norm[..., 1] = yn.astype(np.float32)
    for f in range(F):
        uv_norm[f, int(n_meas[f]):] = 0.0

    # ---- linearisation points: GN triangulation with the *estimated* poses ----------------------------
    R_est = np.array([quat_2_rot(q) for q in clone_q])
    uv_dense = np.zeros((F, C, 2))
    for f in range(F):
        m = int(n_meas[f])
        uv_dense[f, start[f] : start[f] + m] = uv[f, :m].astype(np.float64)
    p0 = p_f + 0.02 * np.linalg.norm(p_f - mid, axis=1, keepdims=True) * frng.standard_normal((F, 3))
    p_FinG = _triangulate_gn(p0, uv_dense, mask, R_est, clone_p, quat_2_rot(calib_q), calib_p, intr, fisheye=fisheye)

    # ---- plane estimates ------------------------------------------------------------------------------
    cp = cp_true + 0.01 * rng.standard_normal(cp_true.shape) if n_planes else np.zeros((0, 3))
    in_state = np.zeros(n_planes, dtype=bool)
    in_state[:n_in_state] = True
    plane_state_id = -np.ones(n_planes, dtype=np.int32)
    for k in range(n_in_state):
        plane_state_id[k] = ids["planes"][k]
        cp[k] = cp_true[k] - err[ids["planes"][k] : ids["planes"][k] + 3]
    cp_fej = cp.copy()
    for k in range(n_in_state):
        cp_fej[k] = cp[k] + 1e-3 * rng.standard_normal(3)

    # SLAM landmarks in state (values only; used by the SLAM-update widening)
    slam_p = np.zeros((n_slam, 3))
    for k in range(n_slam):
        slam_p[k] = mid + np.array([rng.uniform(2, 5), rng.uniform(-2, 2), rng.uniform(-1, 1)])

    sc = Scene(
        C=C,
        F=F,
        N=N,
        ids=ids,
        clone_q=clone_q,
        clone_p=clone_p,
        clone_q_fej=clone_q_fej,
        clone_p_fej=clone_p_fej,
        imu_q=clone_q[-1].copy(),
        imu_p=clone_p[-1].copy(),
        imu_v=np.array([0.0, 0.8, 0.0]),
        imu_bg=1e-3 * rng.standard_normal(3),
        imu_ba=1e-2 * rng.standard_normal(3),
        dt=0.0,
        calib_q=calib_q,
        calib_p=calib_p,
        intr=intr,
        P=P,
        uv=uv,
        uv_norm=uv_norm,
        clone_idx=clone_idx,
        n_meas=n_meas,
        p_FinG=p_FinG,
        plane_id=plane_id,
        cp=cp,
        cp_fej=cp_fej,
        plane_in_state=in_state,
        plane_state_id=plane_state_id,
        slam_p=slam_p,
        fisheye=bool(fisheye),
        opts=dict(
            sigma_px=float(sigma_px),
            sigma_c=float(sigma_c),
            chi2_mult=float(chi2_mult),
            do_fej=bool(do_fej),
            do_calib_pose=bool(calib),
            do_calib_intr=bool(calib),
        ),
        truth=dict(R=R_true, p=p_true, p_f=p_f, cp=cp_true, err=err),
    )
    return sc


def make_stereo_scene(C=8, F=60, seed=0, stereo_frac=0.5, baseline=0.11, **kw):
    """A scene with TWO cameras (update/UpdaterHelper.cpp:335-344 loops over the cameras that measured a feature; state/State.cpp:52-72
    keeps extrinsics and intrinsics per camera).  Built on make_scene: the mono scene's state gets camera 1's 14 calibration columns
    behind camera 0's (the order State::State registers them in: [imu | dt | cam0 extrinsics, intrinsics | cam1 extrinsics,
    intrinsics | clones ...]), a covariance extended with correlated calibration errors for them, and the first
    round(stereo_frac * F) features are ALSO observed by camera 1 at every clone that observes them - their measurement list is
    [camera 0's observations | camera 1's observations], `cam_idx[f, k]` says whose.  Camera 1 sits `baseline` metres beside camera 0
    (along camera 0's x axis) with slightly different intrinsics.
    Extra keys: cam1 = dict(calib_q, calib_p, intr, calib_id, intr_id), cam_idx [F, M] (M = 2 C), ids["calib1"], ids["intr1"]."""
    base = make_scene(C=C, F=F, seed=seed, **kw)
    assert base.cp.shape[0] == 0 and base.opts["do_calib_pose"] and not base.get("fisheye", False)
    rng = np.random.default_rng(9000 + seed)
    N0, at = int(base.N), 30            # camera 1's block goes in front of the clones (id 30 in the mono layout)
    N = N0 + 14
    old_of_new = np.r_[np.arange(at), -np.ones(14, dtype=np.int64), np.arange(at, N0)]
    # covariance: P_ext = B B^T with B = [[L, 0], [M, D]] in the mono order, then permuted: camera 1's calibration errors are
    # correlated with everything else through M
    L = np.linalg.cholesky(base.P)
    scale = np.r_[np.full(3, 3.0e-3), np.full(3, 5.0e-3), np.full(4, 0.5), np.full(4, 2.0e-3)]
    M = 0.3 * scale[:, None] * rng.standard_normal((14, N0)) / np.sqrt(N0)
    B = np.zeros((N, N))
    B[:N0, :N0] = L
    B[N0:, :N0] = M
    B[N0:, N0:] = np.diag(scale)
    Pe = np.zeros((N, N))
    for k in range(N):
        bk = B[:, k]
        nz = np.nonzero(bk)[0]
        Pe[np.ix_(nz, nz)] += np.multiply.outer(bk[nz], bk[nz])
    Pe = 0.5 * (Pe + Pe.T)
    src = np.where(old_of_new >= 0, old_of_new, N0 + (np.arange(N) - at))   # column of Pe (mono order, cam 1 last) for every new column
    P = Pe[np.ix_(src, src)]
    # camera 1: truth beside camera 0's truth, estimate = truth - (its share of the error: drawn, not tied to the mono error draw)
    R_CtoI = T_IMU_CAM[:3, :3]
    p_C0inI = T_IMU_CAM[:3, 3]
    R_ItoC0 = R_CtoI.T
    p_IinC0 = -R_ItoC0 @ p_C0inI
    R_ItoC1_true = rotz(0.01) @ roty(-0.008) @ R_ItoC0
    p_IinC1_true = p_IinC0 - np.array([baseline, 0.0, 0.0])
    intr1_true = INTRINSICS * np.r_[1.01, 1.01, 0.99, 1.01, 1.0, 1.0, 1.0, 1.0]
    e1 = 0.85 * scale * rng.standard_normal(14)
    cam1 = dict(calib_q=quat_boxplus(rot_2_quat(R_ItoC1_true), -e1[:3]), calib_p=p_IinC1_true - e1[3:6], intr=intr1_true - e1[6:],
                calib_id=at, intr_id=at + 6)
    # measurements of camera 1
    n_st = int(round(stereo_frac * F))
    tr = base.truth
    uv1_true, _ = project_all(tr["p_f"], tr["R"], tr["p"], R_ItoC1_true, p_IinC1_true, intr1_true, False)
    Mm = 2 * C
    uv = np.zeros((F, Mm, 2), dtype=np.float32)
    clone_idx = -np.ones((F, Mm), dtype=np.int32)
    cam_idx = np.zeros((F, Mm), dtype=np.int32)
    n_meas = base.n_meas.copy()
    for f in range(F):
        m = int(base.n_meas[f])
        uv[f, :m] = base.uv[f, :m]
        clone_idx[f, :m] = base.clone_idx[f, :m]
        if f < n_st:
            ci = base.clone_idx[f, :m]
            uv[f, m : 2 * m] = (uv1_true[f, ci] + base.opts["sigma_px"] * rng.standard_normal((m, 2))).astype(np.float32)
            clone_idx[f, m : 2 * m] = ci
            cam_idx[f, m : 2 * m] = 1
            n_meas[f] = 2 * m
    ids = dict(base.ids)
    ids["calib1"], ids["intr1"] = at, at + 6
    ids["clones"] = np.asarray(base.ids["clones"]) + 14
    ids["N"] = N
    # normalised measurements as the tracker stores them: undistorted with the estimate of the intrinsics of the camera that took them
    uv_norm = np.zeros((F, Mm, 2), dtype=np.float32)
    for cam, intr_c in ((0, base.intr), (1, cam1["intr"])):
        xn, yn = radtan_undistort(uv[..., 0].astype(np.float64), uv[..., 1].astype(np.float64), intr_c)
        sel = cam_idx == cam
        uv_norm[..., 0][sel] = xn[sel].astype(np.float32)
        uv_norm[..., 1][sel] = yn[sel].astype(np.float32)
    for f in range(F):
        uv_norm[f, int(n_meas[f]):] = 0.0
    sc = Scene(base)
    sc.update(N=N, ids=ids, P=P, uv=uv, clone_idx=clone_idx, cam_idx=cam_idx, n_meas=n_meas, cam1=cam1, n_stereo=n_st, uv_norm=uv_norm)
    return sc


def make_slam_scene(C=11, n_slam=12, seed=0, n_planes=0, ragged=True, outliers=0, wrong_plane=0, **kw):
    """Scene whose F = n_slam features are new observations of SLAM landmarks that are already in the state.

    Landmark f sits at state id ids["slam"][f]; its estimate is truth - err[id:id+3] (consistent with P), its first
    estimate is the value plus a small perturbation.  With n_planes > 0 every plane is in the state and the landmarks are
    spread over the planes (plane_id[f] > 0 for all of them).  The last `outliers` landmarks get gross pixel noise (the
    chi2 gate rejects them with and without the plane rows); the first `wrong_plane` landmarks are associated with a plane
    they do not lie on (the plane rows fail the gate, the no-plane fallback of UpdaterSLAM.cpp:547-609 passes).
    """
    fpp = (n_slam + max(n_planes, 1) - 1) // max(n_planes, 1)
    sc = make_scene(C=C, F=n_slam, seed=seed, ragged=ragged, n_slam=n_slam, n_planes=n_planes, feats_per_plane=fpp,
                    planes_in_state_frac=1.0, min_meas=min(3, C), **kw)
    rng = np.random.default_rng(77 + seed)
    err = sc.truth["err"]
    p = np.zeros((n_slam, 3))
    for f in range(n_slam):
        i = sc.ids["slam"][f]
        p[f] = sc.truth["p_f"][f] - err[i : i + 3]
    sc["p_FinG"] = p
    sc["p_FinG_fej"] = p + 1e-3 * rng.standard_normal(p.shape)
    sc["slam_p"] = p.copy()
    sc["lm_id"] = np.asarray(sc.ids["slam"], dtype=np.int32)
    for f in range(min(outliers, n_slam)):
        m = int(sc.n_meas[n_slam - 1 - f])
        sc.uv[n_slam - 1 - f, :m] += (25.0 * rng.standard_normal((m, 2))).astype(np.float32)
    if n_planes > 1:
        for f in range(min(wrong_plane, n_slam)):
            sc.plane_id[f] = 1 + (int(sc.plane_id[f]) % n_planes)
    return sc


def _rotz(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])


def _roty(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0, s], [0, 1.0, 0], [-s, 0, c]])


def make_imu_scenario(seed=0, rate=400.0, t_state=100.0, dt_cam=0.1, t_off=0.004, fej_perturb=1e-3, low_rate=False,
                      stationary=False, n_cam=1):
    """IMU state + a stream of inertial readings around one camera interval (imitating Simulator's 400 Hz IMU / 10 Hz
    camera, config/sim/estimator_config.yaml:177-178).  Returns (x, imu, time0, time1):
      x    dict q p v bg ba (+ *_fej first estimates)
      imu  [n,7] rows (t, wm xyz, am xyz); readings neither start nor end on time0/time1, so both ends are interpolated
      time0 = t_state + t_off, time1 = t_state + dt_cam + t_off  (Propagator.cpp:67-68)
    `low_rate` produces a stream slower than the camera (exercises the CASE 3.1 branch of select_imu_readings).
    `stationary`: the platform stands still (angular velocity = 0, specific force = R g, velocity estimate ~ 0): the input of
    UpdaterZeroVelocity; `n_cam` camera intervals are covered by the stream.
    """
    rng = np.random.default_rng(4242 + seed)
    q = rot_2_quat(_rotz(0.3 * rng.standard_normal()) @ _roty(0.1 * rng.standard_normal()))
    p = rng.uniform(-2, 2, 3)
    v = 2e-3 * rng.standard_normal(3) if stationary else np.array([0.8, 0.2, -0.05]) + 0.1 * rng.standard_normal(3)
    x = dict(q=q, p=p, v=v, bg=1e-3 * rng.standard_normal(3), ba=1e-2 * rng.standard_normal(3))
    x["q_fej"] = quat_boxplus(x["q"], fej_perturb * rng.standard_normal(3))
    x["p_fej"] = x["p"] + fej_perturb * rng.standard_normal(3)
    x["v_fej"] = x["v"] + fej_perturb * rng.standard_normal(3)
    x["bg_fej"] = x["bg"].copy()
    x["ba_fej"] = x["ba"].copy()
    time0, time1 = t_state + t_off, t_state + dt_cam + t_off
    if low_rate:
        ts = time0 - 0.013 + np.arange(0, 6) * 0.07
    else:
        ts = time0 - 0.0113 + np.arange(0, int((n_cam * dt_cam + 0.03) * rate)) / rate
    R = quat_2_rot(x["q"])
    ph = rng.uniform(0, 2 * np.pi, 6)
    imu = np.zeros((len(ts), 7))
    imu[:, 0] = ts
    for k in range(3):
        amp = 0.0 if stationary else 1.0
        imu[:, 1 + k] = amp * 0.35 * np.sin(2 * np.pi * 0.7 * (ts - ts[0]) + ph[k]) + x["bg"][k]
        imu[:, 4 + k] = amp * 0.6 * np.sin(2 * np.pi * 1.1 * (ts - ts[0]) + ph[3 + k]) + x["ba"][k]
    imu[:, 4:7] += R @ np.array([0, 0, 9.81])
    imu[:, 1:4] += 1.7e-4 * np.sqrt(rate) * rng.standard_normal((len(ts), 3))
    imu[:, 4:7] += 2.0e-3 * np.sqrt(rate) * rng.standard_normal((len(ts), 3))
    return x, imu, time0, time1


PROP_OPTS = dict(sigma_w=1.6968e-04, sigma_a=2.0000e-3, sigma_wb=1.9393e-05, sigma_ab=3.0000e-03, gravity_mag=9.81,
                 use_rk4=True, imu_avg=False, do_fej=True)


def make_planefit_problem(seed=0, n_feats=24, n_obs=9, n_slam=0, ragged=True, outliers=0, px_noise=0.25, sigma_px=1.0,
                          focal=458.0, pt_noise=0.02, cp_noise=0.01, fix_plane=False, sigma_c=0.05):
    """Inputs of one PlaneFitting::plane_fitting / optimize_plane call (track_plane/PlaneFitting.cpp:84-514): points of one
    plane seen from a short camera trajectory.  Camera poses are the clonesCAM entries (R_GtoC, p_CinG); uv_norm are
    normalised image coordinates (Feature::uvs_norm); features without observations play the SLAM features.
    """
    rng = np.random.default_rng(seed)
    # plane: unit normal roughly facing the cameras, 3-4 m away
    nrm = np.array([0.2, 0.1, 1.0]) + 0.2 * rng.standard_normal(3)
    nrm /= np.linalg.norm(nrm)
    dist = 3.5 + 0.5 * rng.random()
    cp_true = nrm * dist
    # orthonormal basis of the plane
    a = np.cross(nrm, [1.0, 0.0, 0.0])
    a /= np.linalg.norm(a)
    b = np.cross(nrm, a)
    nf = n_feats
    st = rng.uniform(-1.2, 1.2, size=(nf, 2))
    p_true = cp_true[None, :] + st[:, :1] * a[None, :] + st[:, 1:] * b[None, :]
    for k in range(outliers):
        p_true[nf - 1 - k] += nrm * (0.25 + 0.1 * k) * (1 if k % 2 else -1)
    # cameras: looking along +z of the global frame with small rotations, moving sideways
    Rs, ps = [], []
    for k in range(n_obs):
        ang = 0.03 * (k - n_obs / 2) + 0.01 * rng.standard_normal()
        R = rotz(0.02 * rng.standard_normal()) @ roty(ang) @ rotx(0.01 * rng.standard_normal())
        Rs.append(R)
        ps.append(np.array([0.08 * k - 0.3, 0.02 * np.sin(k), 0.01 * k]) + 0.01 * rng.standard_normal(3))
    Rs, ps = np.array(Rs), np.array(ps)
    n_obs_f = np.full(nf, n_obs, dtype=np.int32)
    if ragged:
        n_obs_f = rng.integers(max(3, n_obs // 2), n_obs + 1, size=nf).astype(np.int32)
    if n_slam:
        n_obs_f[:n_slam] = 0
    obs_start = np.zeros(nf, dtype=np.int32)
    obs_start[1:] = np.cumsum(n_obs_f)[:-1]
    tot = int(n_obs_f.sum())
    uv = np.zeros((tot, 2))
    Ro = np.zeros((tot, 9))
    po = np.zeros((tot, 3))
    # the tracker is better than the sigma the filter assumes (whitened residuals << 1): with residuals at the Cauchy
    # scale the reweighted Gauss-Newton iteration converges too slowly for the 12 iterations optimize_plane allows
    sig_n = sigma_px / focal
    act_n = px_noise / focal
    for f in range(nf):
        first = int(rng.integers(0, n_obs - n_obs_f[f] + 1)) if n_obs_f[f] else 0
        for k in range(n_obs_f[f]):
            c = first + k
            pc = Rs[c] @ (p_true[f] - ps[c])
            o = obs_start[f] + k
            uv[o] = np.float32(pc[:2] / pc[2] + act_n * rng.standard_normal(2))  # stored as f32 in the reference
            Ro[o] = Rs[c].reshape(-1)
            po[o] = ps[c]
    # estimates as a triangulation leaves them: mostly wrong along the viewing ray, little across it
    ray = p_true - ps[n_obs // 2][None, :]
    ray /= np.linalg.norm(ray, axis=1, keepdims=True)
    p0 = p_true + pt_noise * rng.standard_normal((nf, 1)) * ray + 0.1 * pt_noise * rng.standard_normal((nf, 3))
    if n_slam:
        p0[:n_slam] = p_true[:n_slam] + 0.005 * rng.standard_normal((n_slam, 3))
    cp0 = cp_true + cp_noise * rng.standard_normal(3)
    return Scene(
        n_feats=nf, p_FinG=np.ascontiguousarray(p0), obs_start=obs_start, n_obs=n_obs_f, uv_norm=np.ascontiguousarray(uv),
        R_GtoC=np.ascontiguousarray(Ro), p_CinG=np.ascontiguousarray(po), cp=cp0, cp_true=cp_true, p_true=p_true,
        sigma_px_norm=float(sig_n), sigma_c=float(sigma_c), fix_plane=bool(fix_plane),
        # current IMU pose / extrinsics: identity extrinsics, the last camera
        R_GtoI=Rs[-1].copy(), p_IinG=ps[-1].copy(), R_ItoC=np.eye(3), p_IinC=np.zeros(3),
    )


def slam_rows_on_planes(sc, k_rows, seed=1):
    """SLAM landmarks of a make_scene(n_slam=..., n_planes=...) state placed on its out-of-state planes: the extra rows of
    the MSCKF plane update (update/UpdaterMSCKF.cpp:232-252).  Returns dict(plane [k] 1-based, id [k], p [k,3], p_fej [k,3])."""
    rng = np.random.default_rng(seed)
    out_planes = np.where(~sc.plane_in_state)[0]
    pl = np.array([out_planes[q % len(out_planes)] for q in range(k_rows)], dtype=np.int32)
    p = np.zeros((k_rows, 3))
    truth = sc.get("truth", {})
    for q in range(k_rows):
        cp = truth["cp"][pl[q]] if "cp" in truth else sc.cp[pl[q]]  # the true plane: estimates of it may be centimetres off
        d = np.linalg.norm(cp)
        nrm = cp / d
        x = sc.slam_p[q]
        p[q] = x - (nrm @ x - d) * nrm + 0.01 * rng.standard_normal(3)
    return dict(plane=pl + 1, id=np.asarray(sc.ids["slam"][:k_rows], dtype=np.int32), p=p,
                p_fej=p + 1e-3 * rng.standard_normal((k_rows, 3)))
