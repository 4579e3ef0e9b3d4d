"""TEST INFRASTRUCTURE ONLY - independent numpy restatement of the reference MSCKF(+plane) update path.

This file is the *second* CPU restatement (the first is oracle/ovp_oracle.c).  It exists so the two
can be checked against each other and against finite differences; nothing in the product path
(ov_plane_amd/) may import it.  PARITY UNPINNED: the reference ships no golden vectors for this path
 and cannot be compiled here (needs Eigen/Boost/ov_core), so the pins are
finite-difference Jacobians, algebraic identities and C<->numpy cross-agreement.

Each function cites the reference file:line (relative to /root/reference/ov_plane/src) it follows.
"""
from __future__ import annotations

import numpy as np
from scipy.stats import chi2 as _chi2

from ov_plane_amd.synth import quat_2_rot, quat_boxplus, skew


def chi2_095(k: int) -> float:
    """boost::math::quantile(chi_squared(k), 0.95)  (update/UpdaterMSCKF.cpp:59-62)."""
    return float(_chi2.ppf(0.95, k))


# ------------------------------------------------------------------------------------------------
# camera model: ext ov_core CamRadtan; call sites update/UpdaterHelper.cpp:365,389
# ------------------------------------------------------------------------------------------------
def radtan_distort_d(uv_norm, v):
    fx, fy, cx, cy, k1, k2, p1, p2 = v
    x, y = uv_norm
    r2 = x * x + y * y
    r4 = r2 * r2
    x1 = x * (1 + k1 * r2 + k2 * r4) + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    y1 = y * (1 + k1 * r2 + k2 * r4) + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    return np.array([fx * x1 + cx, fy * y1 + cy])


def radtan_distort_jacobian(uv_norm, v):
    fx, fy, cx, cy, k1, k2, p1, p2 = v
    x, y = uv_norm
    r2 = x * x + y * y
    r4 = r2 * r2
    g = 1 + k1 * r2 + k2 * r4
    x1 = x * g + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    y1 = y * g + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    dz_dzn = np.zeros((2, 2))
    dz_dzn[0, 0] = fx * (g + 2 * k1 * x * x + 4 * k2 * x * x * r2 + 2 * p1 * y + 6 * p2 * x)
    dz_dzn[0, 1] = fx * (2 * k1 * x * y + 4 * k2 * x * y * r2 + 2 * p1 * x + 2 * p2 * y)
    dz_dzn[1, 0] = fy * (2 * k1 * x * y + 4 * k2 * x * y * r2 + 2 * p1 * x + 2 * p2 * y)
    dz_dzn[1, 1] = fy * (g + 2 * k1 * y * y + 4 * k2 * y * y * r2 + 6 * p1 * y + 2 * p2 * x)
    dz_dzeta = np.zeros((2, 8))
    dz_dzeta[0] = [x1, 0, 1, 0, fx * x * r2, fx * x * r4, 2 * fx * x * y, fx * (r2 + 2 * x * x)]
    dz_dzeta[1] = [0, y1, 0, 1, fy * y * r2, fy * y * r4, fy * (r2 + 2 * y * y), 2 * fy * x * y]
    return dz_dzn, dz_dzeta


def equi_distort_d(uv_norm, v):
    """ext ov_core CamEqui::distort_d (fisheye): theta_d = theta + k1 theta^3 + k2 theta^5 + k3 theta^7 + k4 theta^9."""
    fx, fy, cx, cy, k1, k2, k3, k4 = v
    x, y = uv_norm
    r = np.sqrt(x * x + y * y)
    th = np.arctan(r)
    th_d = th + k1 * th**3 + k2 * th**5 + k3 * th**7 + k4 * th**9
    cdist = th_d / r if r > 1e-8 else 1.0
    return np.array([fx * x * cdist + cx, fy * y * cdist + cy])


def equi_distort_jacobian(uv_norm, v):
    """ext CamEqui::compute_distort_jacobian, written as the derivative of cdist(r) * xy instead of the chain the C code uses."""
    fx, fy, cx, cy, k1, k2, k3, k4 = v
    x, y = uv_norm
    r = np.sqrt(x * x + y * y)
    th = np.arctan(r)
    th_d = th + k1 * th**3 + k2 * th**5 + k3 * th**7 + k4 * th**9
    dthd_dr = (1 + 3 * k1 * th**2 + 5 * k2 * th**4 + 7 * k3 * th**6 + 9 * k4 * th**8) / (1 + r * r)
    cdist = th_d / r
    dc_dr = (dthd_dr * r - th_d) / (r * r)
    xy = np.array([x, y])
    d = cdist * np.eye(2) + np.outer(xy, xy) * dc_dr / r
    dz_dzn = np.diag([fx, fy]) @ d
    dz_dzeta = np.zeros((2, 8))
    pw = np.array([th**3, th**5, th**7, th**9])
    dz_dzeta[0] = np.r_[x * cdist, 0, 1, 0, fx * x / r * pw]
    dz_dzeta[1] = np.r_[0, y * cdist, 0, 1, fy * y / r * pw]
    return dz_dzn, dz_dzeta


# ------------------------------------------------------------------------------------------------
# Givens pieces (Eigen JacobiRotation::makeGivens / applyOnTheLeft(0,1,G.adjoint()))
# ------------------------------------------------------------------------------------------------
def make_givens(p, q):
    if q == 0.0:
        return (-1.0 if p < 0 else 1.0), 0.0
    if p == 0.0:
        return 0.0, (1.0 if q < 0 else -1.0)
    if abs(p) > abs(q):
        t = q / p
        u = np.sqrt(1.0 + t * t)
        if p < 0:
            u = -u
        c = 1.0 / u
        s = -t * c
        return c, s
    t = p / q
    u = np.sqrt(1.0 + t * t)
    if q < 0:
        u = -u
    s = -1.0 / u
    c = -t * s
    return c, s


def _rot2(M, r0, c, s, col0=0):
    x = M[r0, col0:].copy()
    y = M[r0 + 1, col0:].copy()
    M[r0, col0:] = c * x - s * y
    M[r0 + 1, col0:] = s * x + c * y


def nullspace_project_inplace(H_f, H_x, res, H_cp=None):
    """update/UpdaterHelper.cpp:515-546 (and the H_cp-carrying twin update/UpdaterPlane.cpp:483-517)."""
    H_f = H_f.copy()
    H_x = H_x.copy()
    res = res.copy().reshape(-1, 1)
    if H_cp is not None:
        H_cp = H_cp.copy()
    nf = H_f.shape[1]
    for n in range(nf):
        for m in range(H_f.shape[0] - 1, n, -1):
            c, s = make_givens(H_f[m - 1, n], H_f[m, n])
            _rot2(H_f, m - 1, c, s, n)
            _rot2(H_x, m - 1, c, s)
            if H_cp is not None:
                _rot2(H_cp, m - 1, c, s)
            _rot2(res, m - 1, c, s)
    if H_cp is not None:
        return H_x[nf:], res[nf:, 0], H_cp[nf:]
    return H_x[nf:], res[nf:, 0]


def measurement_compress_inplace(H_x, res, H_cp=None, use_qr=False):
    """update/UpdaterHelper.cpp:548-579 (twin: update/UpdaterPlane.cpp:519-552).
    use_qr=True swaps the sequential Givens for a Householder QR (same R up to row signs)."""
    if H_x.shape[0] <= H_x.shape[1]:
        return (H_x, res) if H_cp is None else (H_x, res, H_cp)
    r = min(H_x.shape)
    if use_qr:
        Q, R = np.linalg.qr(H_x, mode="reduced")
        out = (R, Q.T @ res)
        if H_cp is not None:
            out = out + (Q.T @ H_cp,)
        return out
    H_x = H_x.copy()
    res = res.copy().reshape(-1, 1)
    if H_cp is not None:
        H_cp = H_cp.copy()
    for n in range(H_x.shape[1]):
        for m in range(H_x.shape[0] - 1, n, -1):
            c, s = make_givens(H_x[m - 1, n], H_x[m, n])
            _rot2(H_x, m - 1, c, s, n)
            if H_cp is not None:
                _rot2(H_cp, m - 1, c, s)
            _rot2(res, m - 1, c, s)
    if H_cp is not None:
        return H_x[:r], res[:r, 0], H_cp[:r]
    return H_x[:r], res[:r, 0]


# ------------------------------------------------------------------------------------------------
# Jacobians: update/UpdaterHelper.cpp:195-513 (GLOBAL_3D only: :39-43; plane rows :448-512)
# ------------------------------------------------------------------------------------------------
def feature_jacobian_full(sc, f, sigma_c=None, p_FinG=None, cp=None, cp_fej=None, plane_state_id=-1, planeid=0,
                          state=None):
    """Returns H_f, H_x, res, order  (order = list of (state_id, size) in the reference's local column order).

    `state` may override the pose tables (dict with clone_q, clone_p, clone_q_fej, clone_p_fej, calib_q, calib_p, intr)."""
    st = sc if state is None else state
    o = sc.opts
    m = int(sc.n_meas[f])
    idx = sc.clone_idx[f, :m]
    ids = sc.ids
    # which camera took measurement k (synth.make_stereo_scene: sc.cam_idx, sc.cam1); UpdaterHelper.cpp:335-344 loops over the
    # cameras of a feature, each with its own extrinsics / intrinsics (State.cpp:52-72)
    cam_of = sc.cam_idx[f, :m] if "cam_idx" in sc else np.zeros(m, dtype=np.int64)
    cams = sorted(set(int(c) for c in cam_of)) or [0]

    def cam_tables(c):
        if c == 0:
            return quat_2_rot(st["calib_q"]), np.asarray(st["calib_p"]), np.asarray(st["intr"]), ids["calib"], ids["intr"]
        c1 = state["cam1"] if (state is not None and "cam1" in state) else sc.cam1
        return quat_2_rot(c1["calib_q"]), np.asarray(c1["calib_p"]), np.asarray(c1["intr"]), ids["calib1"], ids["intr1"]

    # column bookkeeping (UpdaterHelper.cpp:205-277)
    order = []
    col_of = {}
    tot = 0
    for c_ in cams:
        _, _, _, cid_, iid_ = cam_tables(c_)
        if o["do_calib_pose"]:
            col_of[("calib", c_)] = tot
            order.append((int(cid_), 6))
            tot += 6
        if o["do_calib_intr"]:
            col_of[("intr", c_)] = tot
            order.append((int(iid_), 8))
            tot += 8
    for k in range(m):
        ci = int(idx[k])
        if ("c", ci) not in col_of:
            col_of[("c", ci)] = tot
            order.append((int(ids["clones"][ci]), 6))
            tot += 6
    plane_in_state = plane_state_id >= 0
    if planeid != 0 and plane_in_state:
        col_of["plane"] = tot
        order.append((int(plane_state_id), 3))
        tot += 3

    p_f = np.asarray(sc.p_FinG[f] if p_FinG is None else p_FinG, dtype=np.float64)
    p_f_fej = p_f  # MSCKF features: fej = value (UpdaterMSCKF.cpp:499-500,721-722)
    jac = 3 + (3 if (planeid != 0 and not plane_in_state) else 0)
    meas = 3 * m if planeid != 0 else 2 * m
    res = np.zeros(meas)
    H_f = np.zeros((meas, jac))
    H_x = np.zeros((meas, tot))
    white = 1.0 / o["sigma_px"]
    c = 0
    for k in range(m):
        ci = int(idx[k])
        cam = int(cam_of[k])
        R_ItoC, p_IinC, intr, _, _ = cam_tables(cam)
        R_GtoIi = quat_2_rot(st["clone_q"][ci])
        p_IiinG = st["clone_p"][ci]
        p_FinIi = R_GtoIi @ (p_f - p_IiinG)
        p_FinCi = R_ItoC @ p_FinIi + p_IinC
        uv_norm = np.array([p_FinCi[0] / p_FinCi[2], p_FinCi[1] / p_FinCi[2]])
        uv_dist = (equi_distort_d if sc.get("fisheye", False) else radtan_distort_d)(uv_norm, intr)
        uv_m = sc.uv[f, k].astype(np.float64)
        res[c : c + 2] = white * (uv_m - uv_dist)
        if o["do_fej"]:
            R_GtoIi = quat_2_rot(st["clone_q_fej"][ci])
            p_IiinG = st["clone_p_fej"][ci]
            p_FinIi = R_GtoIi @ (p_f_fej - p_IiinG)
            p_FinCi = R_ItoC @ p_FinIi + p_IinC
        dz_dzn, dz_dzeta = (equi_distort_jacobian if sc.get("fisheye", False) else radtan_distort_jacobian)(uv_norm, intr)  # non-FEJ uv_norm (:383,389)
        z = p_FinCi[2]
        dzn_dpfc = np.array([[1 / z, 0, -p_FinCi[0] / (z * z)], [0, 1 / z, -p_FinCi[1] / (z * z)]])
        dpfc_dpfg = R_ItoC @ R_GtoIi
        dpfc_dclone = np.zeros((3, 6))
        dpfc_dclone[:, :3] = R_ItoC @ skew(p_FinIi)
        dpfc_dclone[:, 3:] = -dpfc_dpfg
        dz_dpfc = dz_dzn @ dzn_dpfc
        dz_dpfg = dz_dpfc @ dpfc_dpfg
        H_f[c : c + 2, :3] = white * dz_dpfg
        cc = col_of[("c", ci)]
        H_x[c : c + 2, cc : cc + 6] = white * dz_dpfc @ dpfc_dclone
        if o["do_calib_pose"]:
            dpfc_dcalib = np.zeros((3, 6))
            dpfc_dcalib[:, :3] = skew(p_FinCi - p_IinC)
            dpfc_dcalib[:, 3:] = np.eye(3)
            cc = col_of[("calib", cam)]
            H_x[c : c + 2, cc : cc + 6] += white * dz_dpfc @ dpfc_dcalib
        if o["do_calib_intr"]:
            cc = col_of[("intr", cam)]
            H_x[c : c + 2, cc : cc + 8] = white * dz_dzeta
        c += 2
    if planeid != 0:
        white_c = 1.0 / (o["sigma_c"] if sigma_c is None else sigma_c)
        cp = np.asarray(cp, dtype=np.float64)
        cp_fej = cp if cp_fej is None else np.asarray(cp_fej, dtype=np.float64)
        for _ in range(max(m, 1)):
            d = np.linalg.norm(cp)
            n = cp / d
            res[c] = white_c * (0.0 - (n @ p_f - d))
            lp = p_f
            cpj, dj, nj = cp, d, n
            if o["do_fej"]:
                lp = p_f_fej
                cpj = cp_fej
                dj = np.linalg.norm(cpj)
                nj = cpj / dj
            H_c_plane = white_c * 1.0 / dj * (lp - (nj @ lp) * nj - dj * nj)
            if plane_in_state:
                cc = col_of["plane"]
                H_x[c, cc : cc + 3] = H_c_plane
            else:
                H_f[c, jac - 3 :] = H_c_plane
            H_f[c, :3] = white_c * nj
            c += 1
    return H_f, H_x, res, order


def order_cols(order):
    cols = []
    for sid, sz in order:
        cols.extend(range(sid, sid + sz))
    return np.array(cols, dtype=np.int64)


def get_marginal_covariance(P, order):
    """state/StateHelper.cpp:231-259."""
    cols = order_cols(order)
    return P[np.ix_(cols, cols)].copy()


def ekf_update(P, order, H, res):
    """state/StateHelper.cpp:121-202 with R = I. Returns (P_new, dx)."""
    cols = order_cols(order)
    M_a = P[:, cols] @ H.T
    P_small = P[np.ix_(cols, cols)]
    S = H @ P_small @ H.T + np.eye(H.shape[0])
    S = np.triu(S) + np.triu(S, 1).T
    Sinv = np.linalg.solve(S, np.eye(S.shape[0]))
    Sinv = np.triu(Sinv) + np.triu(Sinv, 1).T
    K = M_a @ Sinv
    Pn = P - K @ M_a.T
    Pn = np.triu(Pn) + np.triu(Pn, 1).T
    dx = K @ res
    return Pn, dx


def ekf_propagation(P, new_start, phi_size, old_order, Phi, Q):
    """state/StateHelper.cpp:41-119. old_order = list of (id,size); Phi [phi_size x sum(old sizes)]."""
    Cov_PhiT = np.zeros((P.shape[0], phi_size))
    loc = 0
    for sid, sz in old_order:
        Cov_PhiT += P[:, sid : sid + sz] @ Phi[:, loc : loc + sz].T
        loc += sz
    Qs = np.triu(Q) + np.triu(Q, 1).T
    PCP = Qs.copy()
    loc = 0
    for sid, sz in old_order:
        PCP += Phi[:, loc : loc + sz] @ Cov_PhiT[sid : sid + sz, :]
        loc += sz
    Pn = P.copy()
    Pn[new_start : new_start + phi_size, :] = Cov_PhiT.T
    Pn[:, new_start : new_start + phi_size] = Cov_PhiT
    Pn[new_start : new_start + phi_size, new_start : new_start + phi_size] = PCP
    return Pn


def apply_dx(sc, dx):
    """ext Type::update per variable: JPL left-multiplicative quats, additive vectors.
    FEJ values are untouched."""
    ids = sc.ids
    out = dict(
        clone_q=sc.clone_q.copy(),
        clone_p=sc.clone_p.copy(),
        clone_q_fej=sc.clone_q_fej,
        clone_p_fej=sc.clone_p_fej,
        calib_q=sc.calib_q.copy(),
        calib_p=sc.calib_p.copy(),
        intr=sc.intr.copy(),
    )
    for i in range(sc.C):
        cid = ids["clones"][i]
        out["clone_q"][i] = quat_boxplus(sc.clone_q[i], dx[cid : cid + 3])
        out["clone_p"][i] = sc.clone_p[i] + dx[cid + 3 : cid + 6]
    out["calib_q"] = quat_boxplus(sc.calib_q, dx[ids["calib"] : ids["calib"] + 3])
    out["calib_p"] = sc.calib_p + dx[ids["calib"] + 3 : ids["calib"] + 6]
    out["intr"] = sc.intr + dx[ids["intr"] : ids["intr"] + 8]
    return out


def msckf_point_update(sc, feats=None, use_qr=False):
    """update/UpdaterMSCKF.cpp:671-814: point-feature loop -> gate -> stack (first-seen order) -> compress -> EKFUpdate.
    Returns dict(dx, P, accepted[F] bool, chi2[F], rows[F], H, res, order)."""
    feats = range(sc.F) if feats is None else feats
    P = sc.P
    Hx_mapping = {}
    order_big = []
    ct_jacob = 0
    blocks = []
    accepted = np.zeros(sc.F, dtype=bool)
    chi2s = np.zeros(sc.F)
    rows = np.zeros(sc.F, dtype=np.int32)
    for f in feats:
        H_f, H_x, res, order = feature_jacobian_full(sc, f)
        H_x, res = nullspace_project_inplace(H_f, H_x, res)
        P_marg = get_marginal_covariance(P, order)
        S = H_x @ P_marg @ H_x.T + np.eye(H_x.shape[0])
        chi2 = float(res @ np.linalg.solve(S, res))
        chi2s[f] = chi2
        rows[f] = res.shape[0]
        if chi2 > sc.opts["chi2_mult"] * chi2_095(res.shape[0]):
            continue
        accepted[f] = True
        for sid, sz in order:
            if sid not in Hx_mapping:
                Hx_mapping[sid] = ct_jacob
                order_big.append((sid, sz))
                ct_jacob += sz
        blocks.append((H_x, res, order))
    ct_meas = sum(b[1].shape[0] for b in blocks)
    if ct_meas < 1:
        return dict(dx=np.zeros(sc.N), P=P.copy(), accepted=accepted, chi2=chi2s, rows=rows)
    Hx_big = np.zeros((ct_meas, ct_jacob))
    res_big = np.zeros(ct_meas)
    r0 = 0
    for H_x, res, order in blocks:
        c0 = 0
        for sid, sz in order:
            Hx_big[r0 : r0 + H_x.shape[0], Hx_mapping[sid] : Hx_mapping[sid] + sz] = H_x[:, c0 : c0 + sz]
            c0 += sz
        res_big[r0 : r0 + res.shape[0]] = res
        r0 += res.shape[0]
    Hc, rc = measurement_compress_inplace(Hx_big, res_big, use_qr=use_qr)
    Pn, dx = ekf_update(P, order_big, Hc, rc)
    return dict(dx=dx, P=Pn, accepted=accepted, chi2=chi2s, rows=rows, H=Hc, res=rc, order=order_big)


# ---- state/Propagator.cpp (a11), independent matrix-form restatement used to cross-check the C oracle ----------------
def _skew(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0.0]])


def _exp_so3(w):
    th = np.linalg.norm(w)
    S = _skew(w)
    if th < 1e-7:
        return np.eye(3) + S + 0.5 * S @ S
    return np.eye(3) + np.sin(th) / th * S + (1 - np.cos(th)) / th**2 * S @ S


def _jl_so3(w):
    th = np.linalg.norm(w)
    if th < 1e-6:
        return np.eye(3)
    a = w / th
    return np.sin(th) / th * np.eye(3) + (1 - np.sin(th) / th) * np.outer(a, a) + (1 - np.cos(th)) / th * _skew(a)


def _Omega(w):
    O = np.zeros((4, 4))
    O[:3, :3] = -_skew(w)
    O[:3, 3] = w
    O[3, :3] = -w
    return O


def _quatnorm(q):
    q = -q if q[3] < 0 else q
    return q / np.linalg.norm(q)


def _quat_mul(q, p):
    from ov_plane_amd.synth import quat_multiply
    return quat_multiply(q, p)


def _q2R(q):
    from ov_plane_amd.synth import quat_2_rot
    return quat_2_rot(q)


def select_imu_readings(imu, t0, t1):
    """Propagator.cpp:227-341 on rows (t, wm, am)."""
    def interp(a, b, t):
        lam = (t - a[0]) / (b[0] - a[0])
        return np.concatenate([[t], (1 - lam) * a[1:] + lam * b[1:]])

    out = []
    n = len(imu)
    for i in range(n - 1):
        a, b = imu[i], imu[i + 1]
        if b[0] > t0 and a[0] < t0:
            out.append(interp(a, b, t0))
            continue
        if a[0] >= t0 and b[0] <= t1:
            out.append(a.copy())
            continue
        if b[0] > t1:
            if a[0] > t1 and i == 0:
                break
            elif a[0] > t1:
                out.append(interp(imu[i - 1], a, t1))
            else:
                out.append(a.copy())
            if out[-1][0] != t1:
                out.append(interp(a, b, t1))
            break
    i = 0
    while i < len(out) - 1:
        if abs(out[i + 1][0] - out[i][0]) < 1e-12:
            out.pop(i)
            continue
        i += 1
    return np.array(out).reshape(-1, 7)


def predict_mean(x, po, dt, w1, a1, w2, a2):
    g = np.array([0, 0, po["gravity_mag"]])
    q0, p0, v0 = x["q"], x["p"], x["v"]
    if not po["use_rk4"]:
        w, a = (0.5 * (w1 + w2), 0.5 * (a1 + a2)) if po["imu_avg"] else (w1, a1)
        wn = np.linalg.norm(w)
        R = _q2R(q0)
        if wn > 1e-20:
            bigO = np.cos(0.5 * wn * dt) * np.eye(4) + np.sin(0.5 * wn * dt) / wn * _Omega(w)
        else:
            bigO = np.eye(4) + 0.5 * dt * _Omega(w)
        return _quatnorm(bigO @ q0), v0 + R.T @ a * dt - g * dt, p0 + v0 * dt + 0.5 * R.T @ a * dt * dt - 0.5 * g * dt * dt
    wal, aj = (w2 - w1) / dt, (a2 - a1) / dt
    dq0 = np.array([0, 0, 0, 1.0])
    ks = []
    dq, vv = dq0, v0
    for s, (frac, adv) in enumerate([(0.0, 0.0), (0.5, 0.5), (0.5, 0.5), (1.0, 1.0)]):
        w, a = w1 + adv * wal * dt, a1 + adv * aj * dt
        if s > 0:
            dq = _quatnorm(dq0 + frac * ks[-1][0])
            vv = v0 + frac * ks[-1][2]
        R = _q2R(_quat_mul(dq, q0))
        ks.append((0.5 * _Omega(w) @ dq * dt, vv * dt, (R.T @ a - g) * dt))
    c = [1 / 6, 1 / 3, 1 / 3, 1 / 6]
    dqf = _quatnorm(dq0 + sum(ci * k[0] for ci, k in zip(c, ks)))
    return _quat_mul(dqf, q0), v0 + sum(ci * k[2] for ci, k in zip(c, ks)), p0 + sum(ci * k[1] for ci, k in zip(c, ks))


def predict_and_compute(x, po, minus, plus):
    """Propagator.cpp:343-454.  Returns (x_new, F, Qd)."""
    dt = plus[0] - minus[0]
    w1, a1 = minus[1:4] - x["bg"], minus[4:7] - x["ba"]
    w2, a2 = plus[1:4] - x["bg"], plus[4:7] - x["ba"]
    nq, nv, npos = predict_mean(x, po, dt, w1, a1, w2, a2)
    g = np.array([0, 0, po["gravity_mag"]])
    F = np.zeros((15, 15))
    G = np.zeros((15, 12))
    th, p, v, bg, ba = slice(0, 3), slice(3, 6), slice(6, 9), slice(9, 12), slice(12, 15)
    Jr = _jl_so3(w1 * dt)   # Jr(-w dt) = Jl(w dt)
    I = np.eye(3)
    if po["do_fej"]:
        Rf = _q2R(x["q_fej"])
        dR = _q2R(nq) @ Rf.T
        F[th, th] = dR
        F[th, bg] = -dR @ Jr * dt
        F[v, th] = -_skew(nv - x["v_fej"] + g * dt) @ Rf.T
        F[v, ba] = -Rf.T * dt
        F[p, th] = -_skew(npos - x["p_fej"] - x["v_fej"] * dt + 0.5 * g * dt * dt) @ Rf.T
        F[p, ba] = -0.5 * Rf.T * dt * dt
        G[th, 0:3] = -dR @ Jr * dt
        G[v, 3:6] = -Rf.T * dt
        G[p, 3:6] = -0.5 * Rf.T * dt * dt
    else:
        R = _q2R(x["q"])
        E = _exp_so3(-w1 * dt)
        F[th, th] = E
        F[th, bg] = -E @ Jr * dt
        F[v, th] = -R.T @ _skew(a1 * dt)
        F[v, ba] = -R.T * dt
        F[p, th] = -0.5 * R.T @ _skew(a1 * dt * dt)
        F[p, ba] = -0.5 * R.T * dt * dt
        G[th, 0:3] = -E @ Jr * dt
        G[v, 3:6] = -R.T * dt
        G[p, 3:6] = -0.5 * R.T * dt * dt
    F[bg, bg] = F[v, v] = F[ba, ba] = F[p, p] = I
    F[p, v] = I * dt
    G[bg, 6:9] = I
    G[ba, 9:12] = I
    Qc = np.diag(np.repeat([po["sigma_w"]**2 / dt, po["sigma_a"]**2 / dt, po["sigma_wb"]**2 * dt, po["sigma_ab"]**2 * dt], 3))
    Qd = G @ Qc @ G.T
    Qd = 0.5 * (Qd + Qd.T)
    xn = dict(x)
    xn.update(q=nq, p=npos, v=nv, q_fej=nq, p_fej=npos, v_fej=nv, bg_fej=x["bg"], ba_fej=x["ba"])
    return xn, F, Qd


def propagate_summed(x, po, imu, t0, t1):
    sel = select_imu_readings(imu, t0, t1)
    Phi, Qs = np.eye(15), np.zeros((15, 15))
    for i in range(len(sel) - 1):
        x, F, Qd = predict_and_compute(x, po, sel[i], sel[i + 1])
        Phi = F @ Phi
        Qs = F @ Qs @ F.T + Qd
        Qs = 0.5 * (Qs + Qs.T)
    last_w = np.zeros(3)
    if len(sel) > 1:
        last_w = sel[-2][1:4] - x["bg"]
    elif len(sel) == 1:
        last_w = sel[-1][1:4] - x["bg"]
    return dict(x=x, Phi=Phi, Q=Qs, last_w=last_w, n_sel=len(sel))


# ---- ext FeatureInitializer, independent matrix-form restatement -----------------------------------------
def triangulate_feature(sc, f, refine=True, opts=None):
    """single_triangulation + single_gaussnewton of feature f of a synth scene (mono).  Returns (ok, p_FinG)."""
    o = dict(max_runs=5, init_lamda=1e-3, max_lamda=1e10, min_dx=1e-6, min_dcost=1e-6, lam_mult=10.0, min_dist=0.10,
             max_dist=60.0, max_baseline=40.0, max_cond_number=10000.0)
    if opts:
        o.update(opts)
    m = int(sc.n_meas[f])
    if m < 2:
        return False, np.zeros(3)
    R_ItoC = _q2R(sc.calib_q)
    cams = []
    for k in range(m):
        ci = int(sc.clone_idx[f, k])
        R_GtoC = R_ItoC @ _q2R(sc.clone_q[ci])
        cams.append((R_GtoC, sc.clone_p[ci] - R_GtoC.T @ sc.calib_p))
    R_GtoA, p_AinG = cams[-1]
    uvn = sc.uv_norm[f, :m]
    A, b, rel = np.zeros((3, 3)), np.zeros(3), []
    for k in range(m):
        R_GtoCi, p_CiinG = cams[k]
        R_AtoCi = R_GtoCi @ R_GtoA.T
        p_CiinA = R_GtoA @ (p_CiinG - p_AinG)
        rel.append((R_AtoCi, p_CiinA, -R_AtoCi @ p_CiinA))
        bi = R_AtoCi.T @ np.array([float(uvn[k, 0]), float(uvn[k, 1]), 1.0])
        bi /= np.linalg.norm(bi)
        Ai = _skew(bi).T @ _skew(bi)
        A += Ai
        b += Ai @ p_CiinA
    pA = np.linalg.solve(A, b)
    sv = np.linalg.svd(A, compute_uv=False)
    if sv[0] / sv[-1] > o["max_cond_number"] or pA[2] < o["min_dist"] or pA[2] > o["max_dist"] or np.isnan(pA).any():
        return False, np.zeros(3)

    def cost(al, be, rho):
        e = 0.0
        for k in range(m):
            R, _, pAC = rel[k]
            h = R @ np.array([al, be, 1.0]) + rho * pAC
            z = np.array([h[0] / h[2], h[1] / h[2]]).astype(np.float32)
            r = uvn[k] - z
            e += float(np.sqrt(r[0] * r[0] + r[1] * r[1])) ** 2
        return e

    if refine:
        rho, al, be = 1.0 / pA[2], pA[0] / pA[2], pA[1] / pA[2]
        lam, eps, runs, recompute = o["init_lamda"], 1e4, 0, True
        cost_old = cost(al, be, rho)
        Hess, grad = np.zeros((3, 3)), np.zeros(3)
        while runs < o["max_runs"] and lam < o["max_lamda"] and eps > o["min_dx"]:
            if recompute:
                Hess[:] = 0
                grad[:] = 0
                for k in range(m):
                    R, _, pAC = rel[k]
                    h = R @ np.array([al, be, 1.0]) + rho * pAC
                    H = np.array([[(R[0, 0] * h[2] - h[0] * R[2, 0]), (R[0, 1] * h[2] - h[0] * R[2, 1]), (pAC[0] * h[2] - h[0] * pAC[2])],
                                  [(R[1, 0] * h[2] - h[1] * R[2, 0]), (R[1, 1] * h[2] - h[1] * R[2, 1]), (pAC[1] * h[2] - h[1] * pAC[2])]]) / h[2]**2
                    z = np.array([h[0] / h[2], h[1] / h[2]]).astype(np.float32)
                    r = (uvn[k] - z).astype(np.float64)
                    grad += H.T @ r
                    Hess += H.T @ H
            Hl = Hess.copy()
            Hl[np.diag_indices(3)] *= (1.0 + lam)
            dx = np.linalg.solve(Hl, grad)
            c = cost(al + dx[0], be + dx[1], rho + dx[2])
            if c <= cost_old and (cost_old - c) / cost_old < o["min_dcost"]:
                al, be, rho = al + dx[0], be + dx[1], rho + dx[2]
                break
            if c <= cost_old:
                recompute, cost_old = True, c
                al, be, rho = al + dx[0], be + dx[1], rho + dx[2]
                runs += 1
                lam /= o["lam_mult"]
                eps = np.linalg.norm(dx)
            else:
                recompute = False
                lam *= o["lam_mult"]
        pA = np.array([al / rho, be / rho, 1.0 / rho])
        u = pA / np.linalg.norm(pA)
        base = max(np.linalg.norm(r[1] - (r[1] @ u) * u) for r in rel)
        if pA[2] < o["min_dist"] or pA[2] > o["max_dist"] or np.linalg.norm(pA) / base > o["max_baseline"]:
            return False, np.zeros(3)
    return True, R_GtoA.T @ pA + p_AinG


# ------------------------------------------------------------------------------------------------
# The point update of UpdaterMSCKF::update (update/UpdaterMSCKF.cpp:695-814) in dense numpy, any number of cameras: rows of every
# feature over all of its measurements, nullspace projection (QR), gate against the prior, stacking, StateHelper::EKFUpdate
# (state/StateHelper.cpp:121-202, R = I).  The reference for scenes the C oracle's record format cannot carry (two cameras).
# ------------------------------------------------------------------------------------------------
def msckf_point_update_dense(sc, chi2_table):
    N = sc.N
    P = np.asarray(sc.P, dtype=np.float64)
    Hs, rs, accepted, chi2 = [], [], np.zeros(sc.F, dtype=bool), np.zeros(sc.F)
    for f in range(sc.F):
        if int(sc.n_meas[f]) < 2:
            continue
        H_f, H_x, res, order = feature_jacobian_full(sc, f)
        Q, _ = np.linalg.qr(H_f, mode="complete")
        Nf = Q[:, H_f.shape[1]:]
        Ho, ro = Nf.T @ H_x, Nf.T @ res
        cols = order_cols(order)
        S = Ho @ P[np.ix_(cols, cols)] @ Ho.T + np.eye(Ho.shape[0])
        chi2[f] = float(ro @ np.linalg.solve(S, ro))
        if chi2[f] > sc.opts["chi2_mult"] * chi2_table[Ho.shape[0]]:
            continue
        accepted[f] = True
        Hb = np.zeros((Ho.shape[0], N))
        Hb[:, cols] = Ho
        Hs.append(Hb)
        rs.append(ro)
    if not Hs:
        return dict(accepted=accepted, chi2=chi2, dx=np.zeros(N), P=P.copy())
    H, r = np.vstack(Hs), np.concatenate(rs)
    S = H @ P @ H.T + np.eye(H.shape[0])
    K = np.linalg.solve(S, H @ P).T
    Pn = P - K @ H @ P
    return dict(accepted=accepted, chi2=chi2, dx=K @ r, P=0.5 * (Pn + Pn.T))
