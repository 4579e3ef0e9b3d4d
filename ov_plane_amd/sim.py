"""Visual-inertial simulator for closed-loop runs of the update path.

Restates what ov_plane's `Simulator` does (sim/Simulator.cpp, sim/SimPlane.h) on top of the SE(3) B-spline of open_vins
(ext ov_core `sim/BsplineSE3`, not in the reference tree: restated from its published algorithm): a trajectory file
(`t tx ty tz qx qy qz qw`, JPL q_GtoI) becomes a uniform cubic B-spline; IMU readings are its angular velocity and specific
force plus bias random walks and white noise, camera frames are projections of a feature map that is grown frame by frame in
front of the camera, half of it on the six faces of a box around the trajectory.

Differences on purpose: the random streams are numpy Generators (the reference draws from std::mt19937 through libstdc++'s
distributions; same seeds do not give the same numbers), one camera.  Everything else follows the reference line by line so the
statistics of what the filter sees are the same: rates, noise scaling, feature-map growth, masks, plane geometry.
"""

import numpy as np

from .synth import INTRINSICS, T_IMU_CAM, quat_2_rot, radtan_distort, radtan_undistort, rot_2_quat, skew

R_ITOC = T_IMU_CAM[:3, :3].T           # kalibr T_imu_cam = T_CtoI
P_IINC = -R_ITOC @ T_IMU_CAM[:3, 3]


# ---- ext quat_ops.h: SO(3) / SE(3) exponentials ---------------------------------------------------------------------
def exp_so3(w):
    th = np.linalg.norm(w)
    S = skew(w)
    if th < 1e-7:
        return np.eye(3) + S
    return np.eye(3) + np.sin(th) / th * S + (1 - np.cos(th)) / th**2 * (S @ S)


def log_so3(R):
    c = 0.5 * (np.trace(R) - 1.0)
    c = min(1.0, max(-1.0, c))
    th = np.arccos(c)
    v = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    if th < 1e-7:
        return 0.5 * v
    return th / (2 * np.sin(th)) * v


def _V(w):
    th = np.linalg.norm(w)
    S = skew(w)
    if th < 1e-7:
        return np.eye(3) + 0.5 * S
    return np.eye(3) + (1 - np.cos(th)) / th**2 * S + (th - np.sin(th)) / th**3 * (S @ S)


def exp_se3(xi):
    T = np.eye(4)
    T[:3, :3] = exp_so3(xi[:3])
    T[:3, 3] = _V(xi[:3]) @ xi[3:]
    return T


def log_se3(T):
    w = log_so3(T[:3, :3])
    return np.concatenate([w, np.linalg.solve(_V(w), T[:3, 3])])


def hat_se3(xi):
    M = np.zeros((4, 4))
    M[:3, :3] = skew(xi[:3])
    M[:3, 3] = xi[3:]
    return M


def inv_se3(T):
    Ti = np.eye(4)
    Ti[:3, :3] = T[:3, :3].T
    Ti[:3, 3] = -T[:3, :3].T @ T[:3, 3]
    return Ti


class BsplineSE3:
    """ext ov_core::BsplineSE3: uniform cubic B-spline over SE(3) control poses T_ItoG sampled from the trajectory."""

    def feed_trajectory(self, traj):
        traj = np.asarray(traj, dtype=float)
        dts = np.diff(traj[:, 0])
        dt = float(np.mean(dts))
        self.dt = 0.05 if dt < 0.05 else dt       # control points no denser than 20 Hz
        poses = []
        for row in traj:
            T = np.eye(4)
            T[:3, :3] = quat_2_rot(row[4:8]).T    # file holds JPL q_GtoI
            T[:3, 3] = row[1:4]
            poses.append(T)
        t_min, t_max = traj[0, 0], traj[-1, 0]
        self.ctrl_t, self.ctrl = [], []
        t, j = t_min, 0
        while True:
            while j + 1 < len(traj) - 1 and traj[j + 1, 0] <= t:
                j += 1
            if t > t_max or j + 1 >= len(traj):
                break
            t0, t1 = traj[j, 0], traj[j + 1, 0]
            lam = (t - t0) / (t1 - t0)
            self.ctrl.append(exp_se3(lam * log_se3(poses[j + 1] @ inv_se3(poses[j]))) @ poses[j])
            self.ctrl_t.append(t)
            t += self.dt
        self.ctrl_t = np.array(self.ctrl_t)
        self.timestamp_start = t_min + 2 * self.dt

    def get_start_time(self):
        return self.timestamp_start

    def _bounding(self, t):
        """control poses 0..3 with t1 <= t < t2, or None outside the spline"""
        i = int(np.searchsorted(self.ctrl_t, t, side="right")) - 1
        if i < 1 or i + 2 >= len(self.ctrl_t):
            return None
        return i

    def _pieces(self, t):
        i = self._bounding(t)
        if i is None:
            return None
        p0, p1, p2, p3 = self.ctrl[i - 1], self.ctrl[i], self.ctrl[i + 1], self.ctrl[i + 2]
        DT = self.ctrl_t[i + 1] - self.ctrl_t[i]
        u = (t - self.ctrl_t[i]) / DT
        w10, w21, w32 = log_se3(inv_se3(p0) @ p1), log_se3(inv_se3(p1) @ p2), log_se3(inv_se3(p2) @ p3)
        b = np.array([(5 + 3 * u - 3 * u**2 + u**3) / 6, (1 + 3 * u + 3 * u**2 - 2 * u**3) / 6, u**3 / 6])
        bd = np.array([(3 - 6 * u + 3 * u**2), (3 + 6 * u - 6 * u**2), 3 * u**2]) / (6 * DT)
        bdd = np.array([(-6 + 6 * u), (6 - 12 * u), 6 * u]) / (6 * DT**2)
        A = [exp_se3(b[0] * w10), exp_se3(b[1] * w21), exp_se3(b[2] * w32)]
        H = [hat_se3(w10), hat_se3(w21), hat_se3(w32)]
        Ad = [bd[k] * H[k] @ A[k] for k in range(3)]
        Add = [bd[k] * H[k] @ Ad[k] + bdd[k] * H[k] @ A[k] for k in range(3)]
        return p0, A, Ad, Add

    def get_pose(self, t):
        pc = self._pieces(t)
        if pc is None:
            return None
        p0, A, _, _ = pc
        T = p0 @ A[0] @ A[1] @ A[2]
        return T[:3, :3].T, T[:3, 3].copy()

    def get_velocity(self, t):
        pc = self._pieces(t)
        if pc is None:
            return None
        p0, A, Ad, _ = pc
        T = p0 @ A[0] @ A[1] @ A[2]
        Td = p0 @ (Ad[0] @ A[1] @ A[2] + A[0] @ Ad[1] @ A[2] + A[0] @ A[1] @ Ad[2])
        W = T[:3, :3].T @ Td[:3, :3]
        w = np.array([W[2, 1], W[0, 2], W[1, 0]])
        return T[:3, :3].T, T[:3, 3].copy(), w, Td[:3, 3].copy()

    def get_acceleration(self, t):
        pc = self._pieces(t)
        if pc is None:
            return None
        p0, A, Ad, Add = pc
        T = p0 @ A[0] @ A[1] @ A[2]
        Td = p0 @ (Ad[0] @ A[1] @ A[2] + A[0] @ Ad[1] @ A[2] + A[0] @ A[1] @ Ad[2])
        Tdd = p0 @ (Add[0] @ A[1] @ A[2] + A[0] @ Add[1] @ A[2] + A[0] @ A[1] @ Add[2] + 2 * Ad[0] @ Ad[1] @ A[2]
                    + 2 * A[0] @ Ad[1] @ Ad[2] + 2 * Ad[0] @ A[1] @ Ad[2])
        R = T[:3, :3]
        W = R.T @ Td[:3, :3]
        w = np.array([W[2, 1], W[0, 2], W[1, 0]])
        Al = R.T @ (Tdd[:3, :3] - Td[:3, :3] @ W)
        alpha = np.array([Al[2, 1], Al[0, 2], Al[1, 0]])
        return R.T, T[:3, 3].copy(), w, Td[:3, 3].copy(), alpha, Tdd[:3, 3].copy()


class SimPlane:
    """sim/SimPlane.h:40-134: a bounded plane from its four corner points, ray intersection, closest-point form."""

    def __init__(self, plane_id, tl, tr, bl, br):
        self.plane_id, self.tl, self.tr, self.bl, self.br = plane_id, tl, tr, bl, br
        N = np.cross(tr - tl, bl - tl)
        self.N = N
        self.D = -N @ tl

    def calculate_intersection(self, origin, bearing):
        den = self.N @ bearing
        if den == 0.0:
            return None
        rng = -(self.N @ origin + self.D) / den
        pt = origin + rng * bearing
        nz = lambda v: v / np.linalg.norm(v)  # noqa: E731
        V1, V2, V3, V4 = nz(self.tr - self.tl), nz(self.bl - self.tl), nz(self.tr - self.br), nz(self.bl - self.br)
        U1, U2 = nz(pt - self.tl), nz(pt - self.br)
        if rng > 0 and U1 @ V1 > 0 and U1 @ V2 > 0 and U2 @ V3 > 0 and U2 @ V4 > 0:
            return rng
        return None

    def cp(self):
        n = self.N / np.linalg.norm(self.N)
        return -self.D / np.linalg.norm(self.N) * n


SIM_DEFAULTS = dict(  # config/sim/estimator_config.yaml:131,172-182 and the IMU noises of kalibr_imu_chain.yaml
    sim_seed_state_init=0, sim_seed_measurements=0, sim_distance_threshold=1.2, sim_freq_cam=10.0, sim_freq_imu=400.0,
    sim_min_feature_gen_distance=2.0, sim_max_feature_gen_distance=5.0, num_pts=250, num_pts_plane=250, gravity_mag=9.81,
    calib_camimu_dt=0.0, sigma_pix=1.0, sigma_w=1.6968e-04, sigma_a=2.0000e-3, sigma_wb=1.9393e-05, sigma_ab=3.0000e-03,
    width=752, height=480)


class Simulator:
    """sim/Simulator.cpp:36-707 for one radtan camera."""

    def __init__(self, traj, **params):
        self.params = dict(SIM_DEFAULTS, **params)
        pr = self.params
        self.intr = np.array(pr.get("intrinsics", INTRINSICS), dtype=float)
        self.R_ItoC = np.array(pr.get("R_ItoC", R_ITOC), dtype=float)
        self.p_IinC = np.array(pr.get("p_IinC", P_IINC), dtype=float)
        traj = np.array(traj, dtype=float)
        traj[:, 3] -= traj[:, 3].mean()                                  # :72-78 average height at z = 0
        self.traj_data = traj
        self.spline = BsplineSE3()
        self.spline.feed_trajectory(traj)
        self.timestamp = self.timestamp_last_imu = self.timestamp_last_cam = self.spline.get_start_time()
        # :97-124 skip ahead until the platform has moved sim_distance_threshold
        _, p_init = self.spline.get_pose(self.timestamp)
        distance = 0.0
        while True:
            pose = self.spline.get_pose(self.timestamp)
            if pose is None:
                raise RuntimeError("[SIM]: unable to find jolt in the groundtruth data to initialize at")
            distance += np.linalg.norm(pose[1] - p_init)
            p_init = pose[1]
            if distance > pr["sim_distance_threshold"]:
                break
            self.timestamp += 1.0 / pr["sim_freq_cam"]
            self.timestamp_last_imu += 1.0 / pr["sim_freq_cam"]
            self.timestamp_last_cam += 1.0 / pr["sim_freq_cam"]
        self.true_bias_gyro, self.true_bias_accel = np.zeros(3), np.zeros(3)
        di = 1.0 / pr["sim_freq_imu"]
        self.hist_true_bias_time = [self.timestamp_last_imu - di, self.timestamp_last_imu, self.timestamp_last_imu + di]
        self.hist_true_bias_gyro = [self.true_bias_gyro.copy() for _ in range(3)]
        self.hist_true_bias_accel = [self.true_bias_accel.copy() for _ in range(3)]
        self.has_skipped_first_bias = False
        self.is_running = True
        self.gen_state_init = np.random.default_rng(pr["sim_seed_state_init"])
        self.gen_meas_imu = np.random.default_rng(pr["sim_seed_measurements"])
        self.gen_meas_cam = np.random.default_rng(pr["sim_seed_measurements"])
        self.planes, self.featmap, self.id_map = [], {}, 0
        self.generate_planes()
        # :190-236 walk along the spline at 4 Hz and top the map up so every view has enough free and planar features
        t = self.spline.get_start_time()
        while True:
            pose = self.spline.get_pose(t)
            if pose is None:
                break
            uvs = self.project_pointcloud(pose[0], pose[1])
            n_free = sum(1 for _, d in uvs if int(d[2]) == -1)
            n_plane = len(uvs) - n_free
            if n_free < pr["num_pts"]:
                self.generate_points(pose[0], pose[1], pr["num_pts"] - n_free, False)
            if n_plane < pr["num_pts_plane"]:
                self.generate_points(pose[0], pose[1], pr["num_pts_plane"] - n_plane, True)
            t += 0.25

    # :277-318
    def get_state(self, t):
        v = self.spline.get_velocity(t)
        ht = self.hist_true_bias_time
        loc = None
        for i in range(len(ht) - 1):
            if ht[i] < t <= ht[i + 1]:
                loc = i
                break
        if v is None or loc is None:
            return None
        lam = (t - ht[loc]) / (ht[loc + 1] - ht[loc])
        bg = (1 - lam) * self.hist_true_bias_gyro[loc] + lam * self.hist_true_bias_gyro[loc + 1]
        ba = (1 - lam) * self.hist_true_bias_accel[loc] + lam * self.hist_true_bias_accel[loc + 1]
        return dict(t=t, q=rot_2_quat(v[0]), p=v[1], v=v[3], bg=bg, ba=ba)

    # :320-381
    def get_next_imu(self):
        pr = self.params
        if self.timestamp_last_cam + 1.0 / pr["sim_freq_cam"] < self.timestamp_last_imu + 1.0 / pr["sim_freq_imu"]:
            return None
        self.timestamp_last_imu += 1.0 / pr["sim_freq_imu"]
        self.timestamp = self.timestamp_last_imu
        acc = self.spline.get_acceleration(self.timestamp)
        if acc is None:
            self.is_running = False
            return None
        R_GtoI, _, w_IinI, _, _, a_IinG = acc
        accel_inI = R_GtoI @ (a_IinG + np.array([0.0, 0.0, pr["gravity_mag"]]))
        dt = 1.0 / pr["sim_freq_imu"]
        g = self.gen_meas_imu
        if self.has_skipped_first_bias:
            self.true_bias_gyro = self.true_bias_gyro + pr["sigma_wb"] * np.sqrt(dt) * g.standard_normal(3)
            self.true_bias_accel = self.true_bias_accel + pr["sigma_ab"] * np.sqrt(dt) * g.standard_normal(3)
            self.hist_true_bias_time.append(self.timestamp_last_imu)
            self.hist_true_bias_gyro.append(self.true_bias_gyro.copy())
            self.hist_true_bias_accel.append(self.true_bias_accel.copy())
        self.has_skipped_first_bias = True
        wm = w_IinI + self.true_bias_gyro + pr["sigma_w"] / np.sqrt(dt) * g.standard_normal(3)
        am = accel_inI + self.true_bias_accel + pr["sigma_a"] / np.sqrt(dt) * g.standard_normal(3)
        return self.timestamp_last_imu, wm, am

    # :383-441
    def get_next_cam(self):
        pr = self.params
        if self.timestamp_last_imu + 1.0 / pr["sim_freq_imu"] < self.timestamp_last_cam + 1.0 / pr["sim_freq_cam"]:
            return None
        self.timestamp_last_cam += 1.0 / pr["sim_freq_cam"]
        self.timestamp = self.timestamp_last_cam
        time_cam = self.timestamp_last_cam - pr["calib_camimu_dt"]
        pose = self.spline.get_pose(self.timestamp)
        if pose is None:
            self.is_running = False
            return None
        uvs = self.project_pointcloud(pose[0], pose[1])
        uvs = uvs[: pr["num_pts"] + pr["num_pts_plane"]]
        out = []
        for fid, d in uvs:
            d = d.copy()
            d[0] += np.float32(pr["sigma_pix"] * self.gen_meas_cam.standard_normal())
            d[1] += np.float32(pr["sigma_pix"] * self.gen_meas_cam.standard_normal())
            out.append((fid, d))
        return time_cam, out

    def _project_all(self, R_GtoI, p_IinG, pts):
        """(:462-487) pixels (f32) of the map points [n,3] and the mask of those that land in the image at a usable depth"""
        pr = self.params
        if len(pts) == 0:
            return np.zeros((0, 2), np.float32), np.zeros(0, bool)
        pc = (pts - p_IinG) @ R_GtoI.T @ self.R_ItoC.T + self.p_IinC
        z = pc[:, 2]
        ok = (z <= pr["sim_max_feature_gen_distance"]) & (z >= 0.1)
        zs = np.where(ok, z, 1.0)
        xn = (pc[:, 0] / zs).astype(np.float32).astype(np.float64)
        yn = (pc[:, 1] / zs).astype(np.float32).astype(np.float64)
        u, v = radtan_distort(xn, yn, self.intr)
        uv = np.stack([u, v], axis=1).astype(np.float32)
        ok &= (uv[:, 0] >= 0) & (uv[:, 0] <= pr["width"]) & (uv[:, 1] >= 0) & (uv[:, 1] <= pr["height"])
        return uv, ok

    def _map_arrays(self):
        ids = np.array(sorted(self.featmap), dtype=np.int64)   # std::map iteration order
        data = np.array([self.featmap[i] for i in ids]) if len(ids) else np.zeros((0, 4))
        return ids, data

    # :443-502 (the 10-pixel occupancy mask keeps one feature per cell, first come first served)
    def project_pointcloud(self, R_GtoI, p_IinG):
        ids, data = self._map_arrays()
        uv, ok = self._project_all(R_GtoI, p_IinG, data[:, :3])
        idx = np.nonzero(ok)[0]
        cell = np.floor(uv[idx, 0] / 10.0).astype(np.int64) * 1000 + np.floor(uv[idx, 1] / 10.0).astype(np.int64)
        _, first = np.unique(cell, return_index=True)
        keep = idx[np.sort(first)]
        return [(int(ids[k]), np.array([uv[k, 0], uv[k, 1], data[k, 3]], dtype=np.float32)) for k in keep]

    # :504-643
    def generate_points(self, R_GtoI, p_IinG, numpts, on_plane):
        pr = self.params
        mask = np.zeros((int(pr["width"] // 10) + 1, int(pr["height"] // 10) + 1), dtype=bool)
        _, data = self._map_arrays()
        pts = data[:, :3].copy()
        uv, ok = self._project_all(R_GtoI, p_IinG, pts)
        mask[np.floor(uv[ok, 0] / 10.0).astype(int), np.floor(uv[ok, 1] / 10.0).astype(int)] = True
        g = self.gen_state_init
        R_CtoG = R_GtoI.T @ self.R_ItoC.T
        i = try_count = 0
        while i < numpts:
            i += 1
            u, v = g.uniform(0, pr["width"]), g.uniform(0, pr["height"])
            count = 0
            while mask[int(np.floor(u / 10.0)), int(np.floor(v / 10.0))]:
                u, v = g.uniform(0, pr["width"]), g.uniform(0, pr["height"])
                count += 1
                if count > 5000:
                    raise RuntimeError("unable to generate feature uv in the mask, are you using too many features???")
            xn, yn = radtan_undistort(np.float32(u), np.float32(v), self.intr)
            bearing = np.array([float(xn), float(yn), 1.0])
            id_plane, depth = -1, np.inf
            if not on_plane:
                depth = g.uniform(pr["sim_min_feature_gen_distance"], pr["sim_max_feature_gen_distance"])
            else:
                origin = p_IinG - R_CtoG @ self.p_IinC
                ray = R_CtoG @ bearing
                for pl in self.planes:
                    rng = pl.calculate_intersection(origin, ray)
                    if rng is not None and rng < depth:
                        depth, id_plane = rng, pl.plane_id
            p_FinC = depth * bearing
            p_FinG = R_GtoI.T @ (self.R_ItoC.T @ (p_FinC - self.p_IinC)) + p_IinG
            closest = np.inf if len(pts) == 0 else np.linalg.norm(pts - p_FinG, axis=1).min()
            if not (0.1 <= p_FinC[2] <= pr["sim_max_feature_gen_distance"]) or closest < 0.10:
                if try_count < 100:
                    i -= 1
                    try_count += 1
                else:
                    try_count = 0
                continue
            try_count = 0
            mask[int(np.floor(u / 10.0)), int(np.floor(v / 10.0))] = True
            self.featmap[self.id_map] = np.array([p_FinG[0], p_FinG[1], p_FinG[2], float(id_plane)])
            pts = np.vstack([pts, p_FinG])
            self.id_map += 1

    # :645-706 six faces of a box around the trajectory
    def generate_planes(self):
        pr = self.params
        pos = self.traj_data[:-1][self.traj_data[:-1, 0] >= self.spline.get_start_time()][:, 1:4]
        mn, mx = pos.min(axis=0), pos.max(axis=0)
        mn[:2] -= 0.7 * pr["sim_min_feature_gen_distance"]
        mn[2] -= 0.24 * pr["sim_min_feature_gen_distance"]
        mx[:2] += 0.7 * pr["sim_min_feature_gen_distance"]
        mx[2] += 0.24 * pr["sim_min_feature_gen_distance"]
        d = mx - mn
        b1 = np.array([mn[0], mn[1], mn[2]])
        b2 = np.array([mn[0] + d[0], mn[1], mn[2]])
        b3 = np.array([mn[0], mn[1] + d[1], mn[2]])
        b4 = np.array([mn[0] + d[0], mn[1] + d[1], mn[2]])
        t1, t2, t3, t4 = (b + np.array([0, 0, d[2]]) for b in (b1, b2, b3, b4))
        corners = [(b1, b2, b3, b4), (t3, t4, t2, t1), (t3, t1, b3, b1), (t1, t2, b1, b2), (t2, t4, b2, b4), (t4, t3, b4, b3)]
        self.planes = [SimPlane(k + 1, *c) for k, c in enumerate(corners)]


def synthetic_trajectory(duration=30.0, rate=100.0, seed=0, pause=0.0):
    """A smooth room-sized loop in the trajectory file format (`t tx ty tz qx qy qz qw`, JPL q_GtoI), standing in for
    ov_data/sim/*.txt (not in the reference tree): ~1 m/s, yaw following the path with roll / pitch wobble.  With `pause`
    the platform stands still for that many seconds first and then eases into the loop (zero-velocity update scenario)."""
    rng = np.random.default_rng(seed)
    ph = rng.uniform(0, 2 * np.pi, 3)
    t_file = np.arange(0.0, duration, 1.0 / rate)
    u = np.maximum(t_file - pause, 0.0)
    t = u * u / (u + 1.5) if pause > 0 else t_file           # path parameter: at rest until `pause`, then accelerating smoothly
    w = 2 * np.pi / 20.0
    p = np.stack([3.0 * np.cos(w * t), 2.0 * np.sin(w * t), 0.4 * np.sin(2 * w * t + ph[0])], axis=1)
    out = np.zeros((len(t), 8))
    out[:, 0] = t_file + 10.0
    out[:, 1:4] = p
    # sensor frame like a forward-looking rig: z_I along the heading, x_I to the right, y_I down (the camera of
    # kalibr_imucam_chain.yaml looks along the IMU's z axis)
    R_ItoB = np.array([[0.0, 0.0, 1.0], [-1.0, 0.0, 0.0], [0.0, -1.0, 0.0]])
    for k, tk in enumerate(t):
        yaw = w * tk + np.pi / 2 + 0.3 * np.sin(0.5 * w * tk + ph[1])
        pitch = 0.15 * np.sin(1.3 * w * tk + ph[2])
        roll = 0.1 * np.sin(0.9 * w * tk)
        cy, sy, cp, sp, cr, sr = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch), np.cos(roll), np.sin(roll)
        R_BtoG = (np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]]) @ np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
                  @ np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]]))
        out[k, 4:8] = rot_2_quat((R_BtoG @ R_ItoB).T)
    return out
