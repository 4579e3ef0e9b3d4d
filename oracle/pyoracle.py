"""TEST INFRASTRUCTURE ONLY - ctypes binding of oracle/libovp_oracle.so (the CPU restatement).

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class OvoOpts(C.Structure):
    _fields_ = [
        ("sigma_px", C.c_double),
        ("chi2_multiplier", C.c_double),
        ("sigma_constraint", C.c_double),
        ("do_fej", C.c_int),
        ("do_calib_camera_pose", C.c_int),
        ("do_calib_camera_intrinsics", C.c_int),
        ("reserved", C.c_int),
    ]


class OvoState(C.Structure):
    _fields_ = [
        ("n_state", C.c_int),
        ("n_clones", C.c_int),
        ("clone_q", C.POINTER(C.c_double)),
        ("clone_p", C.POINTER(C.c_double)),
        ("clone_q_fej", C.POINTER(C.c_double)),
        ("clone_p_fej", C.POINTER(C.c_double)),
        ("clone_id", C.POINTER(C.c_int)),
        ("calib_q", C.c_double * 4),
        ("calib_p", C.c_double * 3),
        ("calib_id", C.c_int),
        ("intrinsics", C.c_double * 8),
        ("intr_id", C.c_int),
        ("cam_fisheye", C.c_int),
    ]


class OvoFeats(C.Structure):
    _fields_ = [
        ("n_feats", C.c_int),
        ("max_meas", C.c_int),
        ("uv", C.POINTER(C.c_float)),
        ("clone_idx", C.POINTER(C.c_int)),
        ("n_meas", C.POINTER(C.c_int)),
        ("p_FinG", C.POINTER(C.c_double)),
        ("p_FinG_fej", C.POINTER(C.c_double)),
    ]


def build(force=False):
    so = os.path.join(_HERE, "libovp_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("ovp_oracle.c", "ovp_oracle.h", "ovp_planefit.c", "ovp_planefit.h", "ovp_oracle_omp.c")]
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def build_fma():
    """Path of the FMA-contracted build of the same source (None when it cannot be built or loaded on this CPU)."""
    so = os.path.join(_HERE, "libovp_oracle_fma.so")
    srcs = [os.path.join(_HERE, f) for f in ("ovp_oracle.c", "ovp_oracle.h", "ovp_planefit.c", "ovp_planefit.h")]
    try:
        if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs):
            subprocess.check_call(["make", "-C", _HERE, "-s", "fma"])
        C.CDLL(so)
    except (OSError, subprocess.CalledProcessError):
        return None
    return so


def build_variant(name):
    """Path of another rounding of the same source (oracle/Makefile: "fma", "x87", "assoc"), None when it cannot be built or
    loaded on this CPU.  The plain build (-ffp-contract off by default at -std=c99) is build()."""
    assert name in ("fma", "x87", "assoc")
    so = os.path.join(_HERE, "libovp_oracle_%s.so" % name)
    srcs = [os.path.join(_HERE, f) for f in ("ovp_oracle.c", "ovp_oracle.h", "ovp_planefit.c", "ovp_planefit.h")]
    try:
        if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs):
            subprocess.check_call(["make", "-C", _HERE, "-s", name])
        C.CDLL(so)
    except (OSError, subprocess.CalledProcessError):
        return None
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libovp_oracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        L.ovo_chi2_quantile_095.restype = C.c_double
        L.ovo_chi2_quantile_095.argtypes = [C.c_int]
        _LIB = L
    return _LIB


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


class Packed:
    """Keeps numpy buffers alive next to the ctypes structs built from a synth.Scene."""

    def __init__(self, sc, feats=None):
        o = sc.opts
        self.opts = OvoOpts(o["sigma_px"], o["chi2_mult"], o["sigma_c"], int(o["do_fej"]), int(o["do_calib_pose"]),
                            int(o["do_calib_intr"]), 0)
        self.clone_q = np.ascontiguousarray(sc.clone_q, dtype=np.float64)
        self.clone_p = np.ascontiguousarray(sc.clone_p, dtype=np.float64)
        self.clone_q_fej = np.ascontiguousarray(sc.clone_q_fej, dtype=np.float64)
        self.clone_p_fej = np.ascontiguousarray(sc.clone_p_fej, dtype=np.float64)
        self.clone_id = np.ascontiguousarray(sc.ids["clones"], dtype=np.int32)
        st = OvoState()
        st.n_state = int(sc.N)
        st.n_clones = int(sc.C)
        st.clone_q = _dp(self.clone_q)
        st.clone_p = _dp(self.clone_p)
        st.clone_q_fej = _dp(self.clone_q_fej)
        st.clone_p_fej = _dp(self.clone_p_fej)
        st.clone_id = _ip(self.clone_id)
        st.calib_q[:] = list(sc.calib_q)
        st.calib_p[:] = list(sc.calib_p)
        st.calib_id = int(sc.ids["calib"]) if o["do_calib_pose"] else -1
        st.intrinsics[:] = list(sc.intr)
        st.intr_id = int(sc.ids["intr"]) if o["do_calib_intr"] else -1
        st.cam_fisheye = 1 if sc.get("fisheye", False) else 0
        self.state = st
        sel = slice(None) if feats is None else np.asarray(feats)
        self.uv = np.ascontiguousarray(sc.uv[sel], dtype=np.float32)
        self.clone_idx = np.ascontiguousarray(sc.clone_idx[sel], dtype=np.int32)
        self.n_meas = np.ascontiguousarray(sc.n_meas[sel], dtype=np.int32)
        self.p_FinG = np.ascontiguousarray(sc.p_FinG[sel], dtype=np.float64)
        fb = OvoFeats()
        fb.n_feats = int(self.uv.shape[0])
        fb.max_meas = int(self.uv.shape[1])
        fb.uv = self.uv.ctypes.data_as(C.POINTER(C.c_float))
        fb.clone_idx = _ip(self.clone_idx)
        fb.n_meas = _ip(self.n_meas)
        fb.p_FinG = _dp(self.p_FinG)
        if "p_FinG_fej" in sc and sc["p_FinG_fej"] is not None:
            self.p_FinG_fej = np.ascontiguousarray(sc["p_FinG_fej"][sel], dtype=np.float64)
            fb.p_FinG_fej = _dp(self.p_FinG_fej)
        self.feats = fb


def feature_jacobian_full(sc, f, planeid=0, cp=None, cp_fej=None, plane_state_id=-1, sigma_c=None):
    pk = Packed(sc)
    m = int(sc.n_meas[f])
    maxr, maxc = 3 * m + 1, 6 * m + 17
    H_f = np.zeros(maxr * 6)
    H_x = np.zeros(maxr * maxc)
    res = np.zeros(maxr)
    rows, cols, hfc, no = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    oid = np.zeros(m + 4, dtype=np.int32)
    osz = np.zeros(m + 4, dtype=np.int32)
    cpa = np.ascontiguousarray(cp if cp is not None else np.zeros(3), dtype=np.float64)
    cpf = np.ascontiguousarray(cp_fej if cp_fej is not None else cpa, dtype=np.float64)
    sc_ = float(sc.opts["sigma_c"] if sigma_c is None else sigma_c)
    lib().ovo_feature_jacobian_full(C.byref(pk.opts), C.byref(pk.state), C.byref(pk.feats), C.c_int(f),
                                    C.c_double(sc_), C.c_int(planeid), _dp(cpa), _dp(cpf), C.c_int(plane_state_id),
                                    _dp(H_f), _dp(H_x), _dp(res), C.byref(rows), C.byref(cols), C.byref(hfc),
                                    _ip(oid), _ip(osz), C.byref(no))
    r, c, h, n = rows.value, cols.value, hfc.value, no.value
    Hf = H_f[: r * h].reshape(h, r).T.copy()
    Hx = H_x[: r * c].reshape(c, r).T.copy()
    order = [(int(oid[i]), int(osz[i])) for i in range(n)]
    return Hf, Hx, res[:r].copy(), order


def feature_jacobian_full_rep(sc, f, rep, anchor_ci):
    """ovo_feature_jacobian_full_rep: the feature held in ext LandmarkRepresentation `rep` (0..5) anchored in clone slot
    anchor_ci.  Returns (H_f [rows, 3 or 1], H_x, res, order)."""
    pk = Packed(sc)
    m = int(sc.n_meas[f])
    maxr, maxc = 3 * m + 1, 6 * m + 17
    H_f = np.zeros(maxr * 6)
    H_x = np.zeros(maxr * maxc)
    res = np.zeros(maxr)
    rows, cols, hfc, no = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    oid = np.zeros(m + 4, dtype=np.int32)
    osz = np.zeros(m + 4, dtype=np.int32)
    rc = lib().ovo_feature_jacobian_full_rep(C.byref(pk.opts), C.byref(pk.state), C.byref(pk.feats), C.c_int(f), C.c_int(rep),
                                             C.c_int(anchor_ci), _dp(H_f), _dp(H_x), _dp(res), C.byref(rows), C.byref(cols),
                                             C.byref(hfc), _ip(oid), _ip(osz), C.byref(no))
    assert rc == 0
    r, c, h, n = rows.value, cols.value, hfc.value, no.value
    return (H_f[: r * h].reshape(h, r).T.copy(), H_x[: r * c].reshape(c, r).T.copy(), res[:r].copy(),
            [(int(oid[i]), int(osz[i])) for i in range(n)])


def feature_jacobian_representation(sc, rep, p_FinG, anchor_ci):
    """ovo_feature_jacobian_representation: (dpfg_dlambda [3, 3 or 1], H_anc [3,6], H_calib [3,6])."""
    pk = Packed(sc)
    dl, Ha, Hc = np.zeros(9), np.zeros(18), np.zeros(18)
    nl = C.c_int()
    pg = np.ascontiguousarray(p_FinG, dtype=np.float64)
    rc = lib().ovo_feature_jacobian_representation(C.byref(pk.opts), C.byref(pk.state), C.c_int(rep), _dp(pg), C.c_int(anchor_ci),
                                                   _dp(dl), C.byref(nl), _dp(Ha), _dp(Hc))
    assert rc == 0
    return dl[: 3 * nl.value].reshape(3, nl.value).copy(), Ha.reshape(3, 6).copy(), Hc.reshape(3, 6).copy()


def anchor_change(sc, rep, old_ci, new_ci, lm_id, p_FinA, p_FinA_fej):
    """ovo_anchor_change on the scene's state (sc.P must cover the landmark columns).  Returns dict(P, p_FinA, p_FinA_fej)."""
    pk = Packed(sc)
    P = np.asfortranarray(sc.P.copy())
    pa = np.ascontiguousarray(p_FinA, dtype=np.float64)
    pf = np.ascontiguousarray(p_FinA_fej, dtype=np.float64)
    pn, pnf = np.zeros(3), np.zeros(3)
    rc = lib().ovo_anchor_change(C.byref(pk.opts), C.byref(pk.state), C.c_int(rep), C.c_int(old_ci), C.c_int(new_ci),
                                 C.c_int(lm_id), _dp(pa), _dp(pf), _dp(P), _dp(pn), _dp(pnf))
    assert rc == 0, rc
    return dict(P=np.ascontiguousarray(P), p_FinA=pn, p_FinA_fej=pnf)


def msckf_point_update(sc, feats=None):
    """Runs ovo_msckf_point_update. Returns dict(dx, P, accepted, chi2, rows_compressed, timings)."""
    pk = Packed(sc, feats)
    P = np.asfortranarray(sc.P.copy())
    N = sc.N
    F = pk.feats.n_feats
    dx = np.zeros(N)
    acc = np.zeros(F, dtype=np.uint8)
    chi2 = np.zeros(F)
    tim = np.zeros(4)
    rc = lib().ovo_msckf_point_update(C.byref(pk.opts), C.byref(pk.state), C.byref(pk.feats), _dp(P), _dp(dx),
                                      acc.ctypes.data_as(C.POINTER(C.c_uint8)), _dp(chi2), _dp(tim))
    return dict(dx=dx, P=np.ascontiguousarray(P), accepted=acc.astype(bool), chi2=chi2, rows_compressed=rc, timings=tim)


def msckf_point_update_omp(sc, feats=None, threads=0):
    """All-cores variant (ovo_msckf_point_update_omp).  Returns dict(dx, P, accepted, chi2, threads, timings)."""
    pk = Packed(sc, feats)
    P = np.asfortranarray(sc.P.copy())
    N = sc.N
    F = pk.feats.n_feats
    dx = np.zeros(N)
    acc = np.zeros(F, dtype=np.uint8)
    chi2 = np.zeros(F)
    tim = np.zeros(4)
    rc = lib().ovo_msckf_point_update_omp(C.byref(pk.opts), C.byref(pk.state), C.byref(pk.feats), _dp(P), _dp(dx),
                                          acc.ctypes.data_as(C.POINTER(C.c_uint8)), _dp(chi2), _dp(tim), C.c_int(int(threads)))
    return dict(dx=dx, P=np.ascontiguousarray(P), accepted=acc.astype(bool), chi2=chi2, threads=rc, timings=tim)


class OvoStateValues(C.Structure):
    _fields_ = [
        ("clone_q", C.POINTER(C.c_double)),
        ("clone_p", C.POINTER(C.c_double)),
        ("calib_q", C.c_double * 4),
        ("calib_p", C.c_double * 3),
        ("intrinsics", C.c_double * 8),
        ("cp", C.POINTER(C.c_double)),
    ]


def msckf_plane_update(sc, libpath=None, slam=None, force=None):
    """Runs ovo_msckf_plane_update on the scene's planar features.
    slam = dict(plane [k] 1-based, id [k], p [k,3], p_fej [k,3]): SLAM landmarks lying on out-of-state planes.
    force = accept / reject byte per plane imposed instead of the gate's decision (diagnostic hook of the oracle).
    Returns dict(P, state values after the plane loop, used[F], plane_ok, plane_chi2, plane_rows[, slam_p])."""
    L = lib() if libpath is None else C.CDLL(libpath)
    if force is not None:
        force = np.ascontiguousarray(force, dtype=np.uint8)
        L.ovo_set_plane_force(force.ctypes.data_as(C.POINTER(C.c_uint8)))
    try:
        return _msckf_plane_update(L, sc, slam)
    finally:
        if force is not None:
            L.ovo_set_plane_force(None)


def _msckf_plane_update(L, sc, slam):
    pk = Packed(sc)
    n_planes = int(sc.cp.shape[0])
    P = np.asfortranarray(sc.P.copy())
    cq = np.ascontiguousarray(sc.clone_q.copy())
    cpv = np.ascontiguousarray(sc.clone_p.copy())
    cp = np.ascontiguousarray(sc.cp.copy())
    cp_in = np.ascontiguousarray(sc.cp.copy())
    cp_fej = np.ascontiguousarray(sc.cp_fej.copy())
    psid = np.ascontiguousarray(sc.plane_state_id, dtype=np.int32)
    pof = np.ascontiguousarray(sc.plane_id, dtype=np.int32)
    val = OvoStateValues()
    val.clone_q = _dp(cq)
    val.clone_p = _dp(cpv)
    val.calib_q[:] = list(sc.calib_q)
    val.calib_p[:] = list(sc.calib_p)
    val.intrinsics[:] = list(sc.intr)
    val.cp = _dp(cp)
    used = np.zeros(sc.F, dtype=np.uint8)
    ok = np.zeros(max(n_planes, 1), dtype=np.uint8)
    chi2 = np.zeros(max(n_planes, 1))
    rows = np.zeros(max(n_planes, 1), dtype=np.int32)
    u8 = C.POINTER(C.c_uint8)
    ns = 0 if slam is None else len(slam["id"])
    s_pl = np.ascontiguousarray(slam["plane"] if ns else [0], dtype=np.int32)
    s_id = np.ascontiguousarray(slam["id"] if ns else [0], dtype=np.int32)
    s_p = np.ascontiguousarray(np.array(slam["p"], dtype=np.float64).copy() if ns else np.zeros((1, 3)))
    s_pf = np.ascontiguousarray(np.array(slam["p_fej"], dtype=np.float64) if ns else np.zeros((1, 3)))
    L.ovo_msckf_plane_update_slam(C.byref(pk.opts), C.byref(pk.state), C.byref(pk.feats), _ip(pof), C.c_int(n_planes),
                                  _dp(cp_in), _dp(cp_fej), _ip(psid), _dp(P), C.byref(val), used.ctypes.data_as(u8),
                                  ok.ctypes.data_as(u8), _dp(chi2), _ip(rows), C.c_int(ns), _ip(s_pl), _ip(s_id), _dp(s_p),
                                  _dp(s_pf))
    out = dict(P=np.ascontiguousarray(P), clone_q=cq, clone_p=cpv, calib_q=np.array(val.calib_q[:]),
               calib_p=np.array(val.calib_p[:]), intr=np.array(val.intrinsics[:]), cp=cp, used=used.astype(bool),
               plane_ok=ok[:n_planes].astype(bool), plane_chi2=chi2[:n_planes], plane_rows=rows[:n_planes])
    if ns:
        out["slam_p"] = s_p
    return out


def initialize(P, order, H_R, H_L, res, r_iso, chi2_mult, do_update=True):
    """ovo_initialize. order = list of (id, size). Returns dict(ok, P (n+k), new_delta[k], dx[n+k], chi2, dof)."""
    n = P.shape[0]
    rows, k = H_L.shape
    cap = n + k
    Pc = np.zeros((cap, cap), order="F")
    Pc[:n, :n] = P
    oid = np.ascontiguousarray([o[0] for o in order], dtype=np.int32)
    osz = np.ascontiguousarray([o[1] for o in order], dtype=np.int32)
    HR = np.asfortranarray(H_R, dtype=np.float64).copy(order="F")
    HL = np.asfortranarray(H_L, dtype=np.float64).copy(order="F")
    r = np.ascontiguousarray(res, dtype=np.float64).copy()
    nn = C.c_int(n)
    nd = np.zeros(k)
    dx = np.zeros(cap)
    chi2 = C.c_double(0)
    dof = C.c_int(0)
    L = lib()
    L.ovo_initialize.restype = C.c_int
    rc = L.ovo_initialize(_dp(Pc), C.c_int(cap), C.byref(nn), _ip(oid), _ip(osz), C.c_int(len(order)), _dp(HR), _dp(HL),
                          C.c_int(rows), C.c_int(k), C.c_double(r_iso), _dp(r), C.c_double(chi2_mult), C.c_int(int(do_update)),
                          _dp(nd), _dp(dx), C.byref(chi2), C.byref(dof))
    n2 = nn.value
    return dict(ok=rc, P=np.ascontiguousarray(Pc[:n2, :n2]), new_delta=nd, dx=dx[:n2], chi2=chi2.value, dof=dof.value)


def plane_init(sc, const_init_multi=5.0, const_init_chi2=1.0, n_extra=None):
    """ovo_plane_init on every plane of the scene that is NOT in the state (scene must have planes_in_state_frac=0)."""
    L = lib()
    pk = Packed(sc)
    n_planes = int(sc.cp.shape[0])
    cap = sc.N + 3 * n_planes
    P = np.zeros((cap, cap), order="F")
    P[: sc.N, : sc.N] = sc.P
    cq = np.ascontiguousarray(sc.clone_q.copy())
    cpv = np.ascontiguousarray(sc.clone_p.copy())
    cpdummy = np.zeros((max(n_planes, 1), 3))
    val = OvoStateValues()
    val.clone_q = _dp(cq)
    val.clone_p = _dp(cpv)
    val.calib_q[:] = list(sc.calib_q)
    val.calib_p[:] = list(sc.calib_p)
    val.intrinsics[:] = list(sc.intr)
    val.cp = _dp(cpdummy)
    cp_in = np.ascontiguousarray(sc.cp, dtype=np.float64)
    pof = np.ascontiguousarray(sc.plane_id, dtype=np.int32)
    used = np.zeros(sc.F, dtype=np.uint8)
    ok = np.zeros(max(n_planes, 1), dtype=np.uint8)
    chi2 = np.zeros(max(n_planes, 1))
    dof = np.zeros(max(n_planes, 1), dtype=np.int32)
    nid = np.zeros(max(n_planes, 1), dtype=np.int32)
    cp_out = np.zeros((max(n_planes, 1), 3))
    nn = C.c_int(sc.N)
    u8 = C.POINTER(C.c_uint8)
    L.ovo_plane_init(C.byref(pk.opts), C.byref(pk.state), C.byref(pk.feats), _ip(pof), C.c_int(n_planes), _dp(cp_in),
                     C.c_double(const_init_multi), C.c_double(const_init_chi2), _dp(P), C.c_int(cap), C.byref(nn), C.byref(val),
                     used.ctypes.data_as(u8), ok.ctypes.data_as(u8), _dp(chi2), _ip(dof), _ip(nid), _dp(cp_out))
    n2 = nn.value
    return dict(P=np.ascontiguousarray(P[:n2, :n2]), n=n2, clone_q=cq, clone_p=cpv, calib_q=np.array(val.calib_q[:]),
                calib_p=np.array(val.calib_p[:]), intr=np.array(val.intrinsics[:]), used=used.astype(bool),
                plane_ok=ok[:n_planes].astype(bool), plane_chi2=chi2[:n_planes], plane_dof=dof[:n_planes],
                new_id=nid[:n_planes], cp=cp_out[:n_planes])


def slam_update(sc, lm_id, use_planes=False, rep=None, anchor=None):
    """ovo_slam_update: features of the scene are observations of landmarks already in the state (ids lm_id[f]).
    rep / anchor [F]: ext LandmarkRepresentation and anchor clone slot of every landmark (None = GLOBAL_3D)."""
    L = lib()
    pk = Packed(sc)
    P = np.asfortranarray(sc.P.copy())
    dx = np.zeros(sc.N)
    acc = np.zeros(sc.F, dtype=np.uint8)
    chi2 = np.zeros(sc.F)
    fb = np.zeros(sc.F, dtype=np.uint8)
    lm = np.ascontiguousarray(lm_id, dtype=np.int32)
    u8 = C.POINTER(C.c_uint8)
    npl = int(sc.cp.shape[0]) if use_planes else 0
    pof = np.ascontiguousarray(sc.plane_id, dtype=np.int32)
    cp = np.ascontiguousarray(sc.cp if npl else np.zeros((1, 3)), dtype=np.float64)
    cpf = np.ascontiguousarray(sc.cp_fej if npl else np.zeros((1, 3)), dtype=np.float64)
    sid = np.ascontiguousarray(sc.plane_state_id if npl else -np.ones(1), dtype=np.int32)
    if rep is None:
        rc = L.ovo_slam_update(C.byref(pk.opts), C.byref(pk.state), C.byref(pk.feats), _ip(lm), _ip(pof), C.c_int(npl), _dp(cp),
                               _dp(cpf), _ip(sid), _dp(P), _dp(dx), acc.ctypes.data_as(u8), _dp(chi2), fb.ctypes.data_as(u8))
    else:
        rp = np.ascontiguousarray(rep, dtype=np.int32)
        an = np.ascontiguousarray(anchor, dtype=np.int32)
        rc = L.ovo_slam_update_rep(C.byref(pk.opts), C.byref(pk.state), C.byref(pk.feats), _ip(lm), _ip(pof), C.c_int(npl), _dp(cp),
                                   _dp(cpf), _ip(sid), _dp(P), _dp(dx), acc.ctypes.data_as(u8), _dp(chi2),
                                   fb.ctypes.data_as(u8), _ip(rp), _ip(an))
    return dict(rc=rc, P=np.ascontiguousarray(P), dx=dx, accepted=acc.astype(bool), chi2=chi2, fellback=fb.astype(bool))


def slam_delayed_init(sc, rep=None):
    """ovo_slam_delayed_init on every feature of the scene (no planes); rep = StateOptions::feat_rep_slam (None = GLOBAL_3D
    through the original entry; out["p"] are the representation parameters)."""
    L = lib()
    pk = Packed(sc)
    cap = sc.N + 3 * sc.F
    P = np.zeros((cap, cap), order="F")
    P[: sc.N, : sc.N] = sc.P
    cq = np.ascontiguousarray(sc.clone_q.copy())
    cpv = np.ascontiguousarray(sc.clone_p.copy())
    dummy = np.zeros((1, 3))
    val = OvoStateValues()
    val.clone_q = _dp(cq)
    val.clone_p = _dp(cpv)
    val.calib_q[:] = list(sc.calib_q)
    val.calib_p[:] = list(sc.calib_p)
    val.intrinsics[:] = list(sc.intr)
    val.cp = _dp(dummy)
    ok = np.zeros(sc.F, dtype=np.uint8)
    chi2 = np.zeros(sc.F)
    nid = np.zeros(sc.F, dtype=np.int32)
    pout = np.zeros((sc.F, 3))
    nn = C.c_int(sc.N)
    if rep is None:
        L.ovo_slam_delayed_init(C.byref(pk.opts), C.byref(pk.state), C.byref(pk.feats), _dp(P), C.c_int(cap), C.byref(nn),
                                C.byref(val), ok.ctypes.data_as(C.POINTER(C.c_uint8)), _dp(chi2), _ip(nid), _dp(pout))
    else:
        rc = L.ovo_slam_delayed_init_rep(C.byref(pk.opts), C.byref(pk.state), C.byref(pk.feats), C.c_int(int(rep)), _dp(P),
                                         C.c_int(cap), C.byref(nn), C.byref(val), ok.ctypes.data_as(C.POINTER(C.c_uint8)),
                                         _dp(chi2), _ip(nid), _dp(pout))
        assert rc == 0, rc
    n2 = nn.value
    return dict(P=np.ascontiguousarray(P[:n2, :n2]), n=n2, clone_q=cq, clone_p=cpv, calib_q=np.array(val.calib_q[:]),
                calib_p=np.array(val.calib_p[:]), intr=np.array(val.intrinsics[:]), ok=ok.astype(bool), chi2=chi2, new_id=nid,
                p=pout)


# ---- state/Propagator.cpp (a11) ----------------------------------------------------------------------------------
class OvoImuState(C.Structure):
    _fields_ = [("q", C.c_double * 4), ("p", C.c_double * 3), ("v", C.c_double * 3), ("bg", C.c_double * 3),
                ("ba", C.c_double * 3), ("q_fej", C.c_double * 4), ("p_fej", C.c_double * 3), ("v_fej", C.c_double * 3),
                ("bg_fej", C.c_double * 3), ("ba_fej", C.c_double * 3)]


class OvoPropOpts(C.Structure):
    _fields_ = [("sigma_w", C.c_double), ("sigma_a", C.c_double), ("sigma_wb", C.c_double), ("sigma_ab", C.c_double),
                ("gravity_mag", C.c_double), ("use_rk4", C.c_int), ("imu_avg", C.c_int), ("do_fej", C.c_int)]


IMU_KEYS = ("q", "p", "v", "bg", "ba", "q_fej", "p_fej", "v_fej", "bg_fej", "ba_fej")


def _imu_struct(x):
    s = OvoImuState()
    for k in IMU_KEYS:
        getattr(s, k)[:] = [float(a) for a in x[k]]
    return s


def _prop_opts(po):
    return OvoPropOpts(po["sigma_w"], po["sigma_a"], po["sigma_wb"], po["sigma_ab"], po["gravity_mag"], int(po["use_rk4"]),
                       int(po["imu_avg"]), int(po["do_fej"]))


def select_imu_readings(imu, time0, time1):
    L = lib()
    imu = np.ascontiguousarray(imu, dtype=np.float64)
    out = np.zeros((imu.shape[0] + 4, 7))
    L.ovo_select_imu_readings.restype = C.c_int
    m = L.ovo_select_imu_readings(_dp(imu), C.c_int(imu.shape[0]), C.c_double(time0), C.c_double(time1), _dp(out),
                                  C.c_int(out.shape[0]))
    return out[:max(m, 0)].copy()


def propagate_summed(x, po, imu, time0, time1):
    """ovo_propagate_summed: x = dict of IMU_KEYS, po = dict of OvoPropOpts fields, imu [n,7] = (t, wm, am)."""
    L = lib()
    s = _imu_struct(x)
    o = _prop_opts(po)
    imu = np.ascontiguousarray(imu, dtype=np.float64)
    Phi = np.zeros((15, 15), order="F")
    Q = np.zeros((15, 15), order="F")
    lw = np.zeros(3)
    n = C.c_int(0)
    L.ovo_propagate_summed.restype = C.c_int
    rc = L.ovo_propagate_summed(C.byref(s), C.byref(o), _dp(imu), C.c_int(imu.shape[0]), C.c_double(time0), C.c_double(time1),
                                _dp(Phi), _dp(Q), _dp(lw), C.byref(n))
    xo = {k: np.array(getattr(s, k)[:]) for k in IMU_KEYS}
    return dict(rc=rc, x=xo, Phi=np.ascontiguousarray(Phi), Q=np.ascontiguousarray(Q), last_w=lw, n_sel=n.value)


def zupt_update(x, po, P, imu, time0, time1, imu_id=0, noise_mult=10.0, chi2_mult=1.0, max_velocity=0.5,
                disparity_passed=False):
    """ovo_zupt_update (update/UpdaterZeroVelocity.cpp:68-318): returns accepted, chi2, rows, dx [n] and the updated P."""
    L = lib()
    s = _imu_struct(x)
    o = _prop_opts(po)
    imu = np.ascontiguousarray(imu, dtype=np.float64)
    n = P.shape[0]
    Pc = np.array(P, order="F", dtype=np.float64)
    dx = np.zeros(n)
    chi2 = C.c_double(0)
    rows = C.c_int(0)
    L.ovo_zupt_update.restype = C.c_int
    rc = L.ovo_zupt_update(C.byref(s), C.byref(o), C.c_int(imu_id), _dp(imu), C.c_int(imu.shape[0]), C.c_double(time0),
                           C.c_double(time1), C.c_double(noise_mult), C.c_double(chi2_mult), C.c_double(max_velocity),
                           C.c_int(int(disparity_passed)), _dp(Pc), C.c_int(n), _dp(dx), C.byref(chi2), C.byref(rows))
    assert rc >= 0, rc
    return dict(accepted=rc == 1, chi2=chi2.value, rows=rows.value, dx=dx, P=np.ascontiguousarray(Pc))


# ---- ext FeatureInitializer ---------------------------------------------------------------------------
class OvoTriangOpts(C.Structure):
    _fields_ = [("refine_features", C.c_int), ("max_runs", C.c_int), ("init_lamda", C.c_double), ("max_lamda", C.c_double),
                ("min_dx", C.c_double), ("min_dcost", C.c_double), ("lam_mult", C.c_double), ("min_dist", C.c_double),
                ("max_dist", C.c_double), ("max_baseline", C.c_double), ("max_cond_number", C.c_double), ("triangulate_1d", C.c_int),
                ("reserved", C.c_int)]


def triang_defaults(**over):
    o = OvoTriangOpts()
    lib().ovo_triang_defaults(C.byref(o))
    for k, v in over.items():
        setattr(o, k, v)
    return o


def triangulate(sc, opts=None, feats=None):
    """ovo_triangulate: single_triangulation + single_gaussnewton of every feature of the scene from sc.uv_norm."""
    L = lib()
    pk = Packed(sc, feats)
    o = opts if opts is not None else triang_defaults()
    sel = np.arange(sc.F) if feats is None else np.asarray(feats)
    uvn = np.ascontiguousarray(sc.uv_norm[sel], dtype=np.float32)
    F = len(sel)
    p = np.zeros((F, 3))
    ok = np.zeros(F, dtype=np.uint8)
    L.ovo_triangulate(C.byref(o), C.byref(pk.state), C.byref(pk.feats), uvn.ctypes.data_as(C.POINTER(C.c_float)), _dp(p),
                      ok.ctypes.data_as(C.POINTER(C.c_uint8)))
    return dict(p_FinG=p, ok=ok.astype(bool))


# ------------------------------------------------------------------------------------------------------------------
# plane fitting (oracle/ovp_planefit.c)
# ------------------------------------------------------------------------------------------------------------------
class OvoMt(C.Structure):
    _fields_ = [("mt", C.c_uint32 * 624), ("idx", C.c_int)]


class OvoPlaneOpt(C.Structure):
    _fields_ = [
        ("n_feats", C.c_int),
        ("p_FinG", C.POINTER(C.c_double)),
        ("obs_start", C.POINTER(C.c_int)),
        ("n_obs", C.POINTER(C.c_int)),
        ("uv_norm", C.POINTER(C.c_double)),
        ("R_GtoC", C.POINTER(C.c_double)),
        ("p_CinG", C.POINTER(C.c_double)),
        ("cp", C.c_double * 3),
        ("sigma_px_norm", C.c_double),
        ("sigma_c", C.c_double),
        ("fix_plane", C.c_int),
        ("R_GtoI", C.c_double * 9),
        ("p_IinG", C.c_double * 3),
        ("R_ItoC", C.c_double * 9),
        ("p_IinC", C.c_double * 3),
    ]


def mt_values(seed, count):
    L = lib()
    L.ovo_mt_next.restype = C.c_uint32
    g = OvoMt()
    L.ovo_mt_seed(C.byref(g), C.c_uint32(seed))
    return [int(L.ovo_mt_next(C.byref(g))) for _ in range(count)]


def shuffles(seed, n, rounds, variant):
    """`rounds` consecutive std::shuffle calls of 0..n-1 with one std::mt19937(seed), as plane_fitting does."""
    L = lib()
    g = OvoMt()
    L.ovo_mt_seed(C.byref(g), C.c_uint32(seed))
    out = []
    for _ in range(rounds):
        v = np.arange(n, dtype=np.int32)
        L.ovo_shuffle(_ip(v), C.c_int(n), C.byref(g), C.c_int(variant))
        out.append(v.copy())
    return np.array(out)


def fit_plane(pts, cond_thresh=1e9, cond_check=True):
    L = lib()
    pts = np.ascontiguousarray(pts, dtype=np.float64)
    abcd = np.zeros(4)
    ok = L.ovo_fit_plane(_dp(pts), C.c_int(len(pts)), C.c_double(cond_thresh), C.c_int(int(cond_check)), _dp(abcd))
    return bool(ok), abcd


def plane_fitting(pts, min_inlier_num, max_cond, variant=0):
    L = lib()
    pts = np.ascontiguousarray(pts, dtype=np.float64)
    n = len(pts)
    abcd = np.zeros(4)
    inl = np.zeros(max(n, 1), dtype=np.uint8)
    cnt = C.c_int(0)
    ok = L.ovo_plane_fitting(_dp(pts), C.c_int(n), C.c_int(min_inlier_num), C.c_double(max_cond), C.c_int(variant), _dp(abcd),
                             inl.ctypes.data_as(C.POINTER(C.c_ubyte)), C.byref(cnt))
    return dict(ok=bool(ok), abcd=abcd, inlier=inl[:n].astype(bool), n_inliers=int(cnt.value))


def _planeopt_struct(pb):
    keep = {k: np.ascontiguousarray(pb[k], dtype=np.float64) for k in ("p_FinG", "uv_norm", "R_GtoC", "p_CinG")}
    keep["obs_start"] = np.ascontiguousarray(pb["obs_start"], dtype=np.int32)
    keep["n_obs"] = np.ascontiguousarray(pb["n_obs"], dtype=np.int32)
    s = OvoPlaneOpt()
    s.n_feats = int(pb["n_feats"])
    s.p_FinG = _dp(keep["p_FinG"])
    s.obs_start = _ip(keep["obs_start"])
    s.n_obs = _ip(keep["n_obs"])
    s.uv_norm = _dp(keep["uv_norm"])
    s.R_GtoC = _dp(keep["R_GtoC"])
    s.p_CinG = _dp(keep["p_CinG"])
    s.cp = (C.c_double * 3)(*np.asarray(pb["cp"], dtype=np.float64))
    s.sigma_px_norm = float(pb["sigma_px_norm"])
    s.sigma_c = float(pb["sigma_c"])
    s.fix_plane = int(bool(pb["fix_plane"]))
    s.R_GtoI = (C.c_double * 9)(*np.asarray(pb["R_GtoI"], dtype=np.float64).reshape(-1))
    s.p_IinG = (C.c_double * 3)(*np.asarray(pb["p_IinG"], dtype=np.float64))
    s.R_ItoC = (C.c_double * 9)(*np.asarray(pb["R_ItoC"], dtype=np.float64).reshape(-1))
    s.p_IinC = (C.c_double * 3)(*np.asarray(pb["p_IinC"], dtype=np.float64))
    return s, keep


def optimize_plane(pb):
    L = lib()
    s, keep = _planeopt_struct(pb)
    nf = int(pb["n_feats"])
    cp = np.zeros(3)
    p = np.zeros((max(nf, 1), 3))
    kept = np.zeros(max(nf, 1), dtype=np.uint8)
    nk, it = C.c_int(0), C.c_int(0)
    ok = L.ovo_optimize_plane(C.byref(s), _dp(cp), _dp(p), kept.ctypes.data_as(C.POINTER(C.c_ubyte)), C.byref(nk), C.byref(it))
    return dict(ok=bool(ok), cp=cp, p_FinG=p[:nf], kept=kept[:nf].astype(bool), n_kept=int(nk.value), iterations=int(it.value))


def planeopt_cost(pb, p_FinG, cp, grad=False):
    L = lib()
    L.ovo_planeopt_cost.restype = C.c_double
    s, keep = _planeopt_struct(pb)
    p_FinG = np.ascontiguousarray(p_FinG, dtype=np.float64)
    cp = np.ascontiguousarray(cp, dtype=np.float64)
    if not grad:
        return float(L.ovo_planeopt_cost(C.byref(s), _dp(p_FinG), _dp(cp), None, None))
    gp = np.zeros_like(p_FinG)
    gc = np.zeros(3)
    c = float(L.ovo_planeopt_cost(C.byref(s), _dp(p_FinG), _dp(cp), _dp(gp), _dp(gc)))
    return c, gp, gc
