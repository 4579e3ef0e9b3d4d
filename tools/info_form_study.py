"""Numerics of the plane loop in accumulated-information form against the T-form, truth = x87 long double."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ov_plane_amd.synth import make_scene
from oracle import np_ref

def chol_ld(A):
    n = A.shape[0]; L = np.zeros_like(A)
    for j in range(n):
        v = A[j:, j] - L[j:, :j] @ L[j, :j]
        L[j:, j] = v / np.sqrt(v[0])
    return L
def solve_lower(L, B):
    n = L.shape[0]; X = np.array(B, dtype=L.dtype, copy=True)
    for i in range(n):
        X[i] = (X[i] - L[i, :i] @ X[:i]) / L[i, i]
    return X
def solve_upper(U, B):
    n = U.shape[0]; X = np.array(B, dtype=U.dtype, copy=True)
    for i in range(n - 1, -1, -1):
        X[i] = (X[i] - U[i, i + 1:] @ X[i + 1:]) / U[i, i]
    return X
def inv_spd(A):
    L = chol_ld(A); Li = solve_lower(L, np.eye(A.shape[0], dtype=A.dtype)); return Li.T @ Li

sc = make_scene(C=30, F=2000, seed=0, n_planes=20, feats_per_plane=50, planes_in_state_frac=0.5, chi2_mult=1.0)
N = sc.N; P0 = sc.P
print("N", N, "cond(P)", np.linalg.cond(P0), "cond(DPD)", np.linalg.cond(P0 / np.sqrt(np.outer(np.diag(P0), np.diag(P0)))))
pairs = []
for k in range(20):
    A = np.zeros((N, N)); b = np.zeros(N)
    for f in range(50 * k, 50 * k + 50):
        H_f, H_x, res, order = np_ref.feature_jacobian_full(sc, f)
        H_x, res = np_ref.nullspace_project_inplace(H_f, H_x, res)
        cols = np_ref.order_cols(order)
        A[np.ix_(cols, cols)] += H_x.T @ H_x; b[cols] += H_x.T @ res
    pairs.append((A, b))
print("|A| max", max(np.abs(a).max() for a, _ in pairs))
LD = np.longdouble
# truth
t0 = time.time()
M = inv_spd(P0.astype(LD)); truth = []
for A, b in pairs:
    M = M + A.astype(LD)
    Pk = inv_spd(M); truth.append((Pk @ b.astype(LD), Pk))
print("truth", time.time() - t0)
def nerr(P, Pt):
    d = np.sqrt(np.abs(np.diag(Pt).astype(np.float64))); return float((np.abs(P - Pt.astype(np.float64)) / np.outer(d, d)).max())
# T-form (f64)
L0 = np.linalg.cholesky(P0); T = np.eye(N); e_t = []
for (A, b), (dxt, Pt) in zip(pairs, truth):
    T = T + L0.T @ (A @ L0); Lt = np.linalg.cholesky(T)
    z = np.linalg.solve(Lt, L0.T @ b); dx = L0 @ np.linalg.solve(Lt.T, z)
    V = np.linalg.solve(Lt, L0.T); e_t.append((np.abs(dx - dxt.astype(np.float64)).max(), nerr(V.T @ V, Pt)))
# information form, Jacobi-scaled by D = sqrt(diag P0): Ms = D P0^-1 D + D A D
D = np.sqrt(np.diag(P0)); Ps = P0 / np.outer(D, D)
Ls = np.linalg.cholesky(Ps); Li = np.linalg.solve(Ls, np.eye(N)); Ms = Li.T @ Li; e_i = []
for (A, b), (dxt, Pt) in zip(pairs, truth):
    Ms = Ms + A * np.outer(D, D); Lm = np.linalg.cholesky(Ms)
    z = np.linalg.solve(Lm, D * b); dx = D * np.linalg.solve(Lm.T, z)
    W = np.linalg.solve(Lm, np.diag(D)); e_i.append((np.abs(dx - dxt.astype(np.float64)).max(), nerr(W.T @ W, Pt)))
print("dx scale", max(np.abs(t[0]).max() for t in truth))
for k in (0, 1, 5, 10, 19):
    print(k, "T-form dx %.2e P %.2e | info dx %.2e P %.2e" % (e_t[k] + e_i[k]))
print("max T-form", max(e[0] for e in e_t), max(e[1] for e in e_t), " info", max(e[0] for e in e_i), max(e[1] for e in e_i))

def run_case(name, P0):
    M = inv_spd(P0.astype(LD)); truth = []
    for A, b in pairs:
        M = M + A.astype(LD); Pk = inv_spd(M); truth.append((Pk @ b.astype(LD), Pk))
    L0 = np.linalg.cholesky(P0); T = np.eye(N); e_t = []
    for (A, b), (dxt, Pt) in zip(pairs, truth):
        T = T + L0.T @ (A @ L0); Lt = np.linalg.cholesky(T)
        z = np.linalg.solve(Lt, L0.T @ b); dx = L0 @ np.linalg.solve(Lt.T, z)
        V = np.linalg.solve(Lt, L0.T); e_t.append((np.abs(dx - dxt.astype(np.float64)).max(), nerr(V.T @ V, Pt)))
    D = np.sqrt(np.diag(P0)); Ps = P0 / np.outer(D, D)
    Ls = np.linalg.cholesky(Ps); Li = np.linalg.solve(Ls, np.eye(N)); Ms = Li.T @ Li; e_i = []
    for (A, b), (dxt, Pt) in zip(pairs, truth):
        Ms = Ms + A * np.outer(D, D); Lm = np.linalg.cholesky(Ms)
        z = np.linalg.solve(Lm, D * b); dx = D * np.linalg.solve(Lm.T, z)
        W = np.linalg.solve(Lm, np.diag(D)); e_i.append((np.abs(dx - dxt.astype(np.float64)).max(), nerr(W.T @ W, Pt)))
    print("%s: cond(P) %.1e cond(DPD) %.1e | T-form dx %.1e P %.1e | info-form dx %.1e P %.1e" % (
        name, np.linalg.cond(P0), np.linalg.cond(Ps), max(e[0] for e in e_t), max(e[1] for e in e_t), max(e[0] for e in e_i), max(e[1] for e in e_i)))

# (1) spectrum of the correlation matrix stretched to cond 1e8
D = np.sqrt(np.diag(P0)); Cn = P0 / np.outer(D, D); w, Q = np.linalg.eigh(Cn)
p = np.log(1e8 / 1e4) / np.log(w.max() / w.min()) + 1.0
for target in (1e6, 1e8):
    p = np.log(target) / np.log(w.max() / w.min())
    C2 = (Q * w ** p) @ Q.T; d2 = np.sqrt(np.diag(C2)); C2 = C2 / np.outer(d2, d2)
    run_case("stretched correlation %.0e" % target, C2 * np.outer(D, D))
# (2) the prior every update of a running filter sees: newest clone = exact copy of the IMU pose (StateHelper.cpp:346-396) with the
# 1e-11 relative inflation of ovp_cov_clone on its diagonal
ids = sc.ids; imu = 0; cl = ids["clones"][-1]
P2 = P0.copy(); src = list(range(imu, imu + 6)); dst = list(range(cl, cl + 6))
P2[dst, :] = P2[src, :]; P2[:, dst] = P2[:, src]
for k in dst: P2[k, k] *= 1.0 + 1e-11
run_case("fresh clone (1e-11 inflation)", P2)
