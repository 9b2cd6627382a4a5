"""TEST INFRASTRUCTURE ONLY -- CPU check for the nominal scheme on rank-deficient (noise-free) data.

The nominal QP of the reference (direct_data_driven_mpc_controller.py:506-538,549-629,679-711):

    min  (ubar_P - 1 x u_s)' R (...) + (ybar_P - 1 x y_s)' Q (...)
    s.t. [ubar; ybar] = H alpha,  internal window = past data,  [TEC] terminal window = setpoint

has a unique [ubar; ybar] but no unique alpha when H is rank-deficient.  This solver works on H itself with
orthogonal factorisations only (SVD of H for a basis of its range, SVD of the constraint block for its null
space, least squares for the rest) -- deliberately a different route from the Gram/Cholesky one the GPU
kernels take.  Parity unpinned against the reference (cvxpy absent), like the rest of the QP oracle.
solve_nominal_exact: diagonal Q, R only; solve_nominal_model_based also takes dense ones."""
from __future__ import annotations

import numpy as np

from .ddmpc_oracle import QPSpec, hankel_matrix


def _svd(a, full_matrices):
    """numpy's divide-and-conquer SVD occasionally fails to converge on these matrices; fall back to the QR-iteration
    driver then."""
    try:
        return np.linalg.svd(a, full_matrices=full_matrices)
    except np.linalg.LinAlgError:
        import scipy.linalg
        return scipy.linalg.svd(a, full_matrices=full_matrices, lapack_driver="gesvd")


def solve_nominal_exact(spec: QPSpec, u_d, y_d, u_past, y_past, rank_tol: float = 1e-10, feas_tol: float = 1e-7):
    """Returns dict(status, optimal_u, cost, residual, rank)."""
    n, m, p, L_ = spec.n, spec.m, spec.p, spec.L
    Ln = L_ + n
    H = np.vstack([hankel_matrix(u_d, Ln), hankel_matrix(y_d, Ln)])              # rows: ubar (k*m+ch), then ybar
    u_past = np.asarray(u_past, float).ravel(); y_past = np.asarray(y_past, float).ravel()
    fixed, weight, target = {}, {}, {}
    for k in range(Ln):
        kp = k - n
        for ch in range(m):
            i = k * m + ch
            if kp < 0:
                fixed[i] = u_past[k * m + ch]                                       # :577-581
            elif spec.tec and kp >= L_ - n:
                fixed[i] = spec.u_s[ch]                                             # :612-627
            else:
                weight[i] = spec.R[kp * m + ch, kp * m + ch]; target[i] = spec.u_s[ch]
        for ch in range(p):
            i = Ln * m + k * p + ch
            if kp < 0:
                fixed[i] = y_past[k * p + ch]
            elif spec.tec and kp >= L_ - n:
                fixed[i] = spec.y_s[ch]
            else:
                weight[i] = spec.Q[kp * p + ch, kp * p + ch]; target[i] = spec.y_s[ch]
    F, R = sorted(fixed), sorted(weight)
    f = np.array([fixed[i] for i in F]); W = np.array([weight[i] for i in R]); zs = np.array([target[i] for i in R])
    U, S, _ = _svd(H, False)
    k = int(np.sum(S > S[0] * rank_tol))
    U = U[:, :k]                                                                    # z = U c spans range(H)
    Uf, Sf, Vft = _svd(U[F], True)
    kf = int(np.sum(Sf > Sf[0] * 1e-9))
    c_p = Vft[:kf].T @ ((Uf[:, :kf].T @ f) / Sf[:kf])                               # least-squares particular solution
    N = Vft[kf:].T                                                                  # null space of the constraint block
    residual = float(np.max(np.abs(U[F] @ c_p - f)))
    A = np.sqrt(W)[:, None] * (U[R] @ N)
    d = np.linalg.lstsq(A, np.sqrt(W) * (zs - U[R] @ c_p), rcond=None)[0]
    z = U @ (c_p + N @ d)
    scale = max(1.0, float(np.max(np.abs(f))))
    status = "optimal" if residual <= feas_tol * scale else "infeasible"
    return dict(status=status, optimal_u=z[:Ln * m][n * m:], cost=float(np.sum(W * (z[R] - zs) ** 2)),
                residual=residual, rank=k)


def solve_nominal_model_based_batch(spec, plant, u_past, y_past):
    """solve_nominal_model_based for a batch of past windows of ONE plant (u_past [B, n*m], y_past [B, n*p]): the basis of the
    trajectory space and its factorisations do not depend on the window, so they are formed once.  Returns
    (optimal_u [B, L*m], cost [B], feas_residual [B]).  TEST INFRASTRUCTURE ONLY."""
    u_past = np.atleast_2d(np.asarray(u_past, float)); y_past = np.atleast_2d(np.asarray(y_past, float))
    B = u_past.shape[0]
    one = solve_nominal_model_based(spec, plant, u_past[0], y_past[0], _parts=True)
    Qb, F, R, W, zs, Uf, Sf, Vft, kf, fmap = (one[k] for k in ("Qb", "F", "R", "W", "zs", "Uf", "Sf", "Vft", "kf", "fmap"))
    f = np.tile(one["f"][:, None], (1, B))                       # fixed values: the setpoint entries are the same, the past entries vary
    for row, (kind, idx) in enumerate(fmap):
        if kind == "u": f[row] = u_past[:, idx]
        elif kind == "y": f[row] = y_past[:, idx]
    c_p = Vft[:kf].T @ ((Uf[:, :kf].T @ f) / Sf[:kf, None]); Nn = Vft[kf:].T
    sw = np.sqrt(W)
    dd = np.linalg.lstsq(sw[:, None] * (Qb[R] @ Nn), sw[:, None] * (zs[:, None] - Qb[R] @ c_p), rcond=None)[0]
    z = Qb @ (c_p + Nn @ dd)
    n, m, Ln = spec.n, spec.m, spec.L + spec.n
    return (z[:Ln * m][n * m:].T.copy(), np.sum(W[:, None] * (z[R] - zs[:, None]) ** 2, axis=0),
            np.max(np.abs(Qb[F] @ c_p - f), axis=0))


def solve_nominal_model_based(spec, plant, u_past, y_past, _parts=False):
    """The nominal QP on EXACT data restated on a basis of the plant's own trajectory space built from (A, B, C)
    (D = 0): every noise-free trajectory of length L+n is [u; y] = M [x_0; u], so range(H) = range(M) whenever the data
    are persistently exciting.  Not data-driven and well conditioned (no Hankel matrix, no Gram matrix): the yardstick
    for the GPU kernels AND for the SVD route above on exact data, where the latter is only ~1e-8 accurate at
    cfg-5 size.  Returns optimal_u = ubar[n*m:] and the cost.  TEST INFRASTRUCTURE ONLY."""
    n, m, p, Lh = spec.n, spec.m, spec.p, spec.L
    Ln = Lh + n
    A_, B_, C_ = (np.asarray(plant[k], float) for k in ("A", "B", "C"))
    ns = A_.shape[0]
    u_s = np.asarray(spec.u_s, float).reshape(-1); y_s = np.asarray(spec.y_s, float).reshape(-1)
    u_past = np.asarray(u_past, float).reshape(-1); y_past = np.asarray(y_past, float).reshape(-1)
    rdiag, qdiag = np.diag(spec.R), np.diag(spec.Q)
    Rm, Qm = np.asarray(spec.R, float), np.asarray(spec.Q, float)
    dense = not (np.array_equal(Rm, np.diag(rdiag)) and np.array_equal(Qm, np.diag(qdiag)))
    M = np.zeros((Ln * (m + p), ns + Ln * m))
    for k in range(Ln):
        M[k * m:(k + 1) * m, ns + k * m: ns + (k + 1) * m] = np.eye(m)
    Ak = np.eye(ns); O = []
    for k in range(Ln):
        O.append(C_ @ Ak); Ak = A_ @ Ak
    for k in range(Ln):
        M[Ln * m + k * p: Ln * m + (k + 1) * p, :ns] = O[k]
        for j in range(k):
            M[Ln * m + k * p: Ln * m + (k + 1) * p, ns + j * m: ns + (j + 1) * m] = O[k - 1 - j] @ B_
    F, R, f, W, zs, fmap, widx = [], [], [], [], [], [], []
    for k in range(Ln):
        kp = k - n
        for ch in range(m):
            i = k * m + ch
            if kp < 0: F.append(i); f.append(u_past[k * m + ch]); fmap.append(("u", k * m + ch))
            elif spec.tec and kp >= Lh - n: F.append(i); f.append(u_s[ch]); fmap.append(("s", 0))
            else: R.append(i); W.append(rdiag[kp * m + ch]); zs.append(u_s[ch]); widx.append(("R", kp * m + ch))
    for k in range(Ln):
        kp = k - n
        for ch in range(p):
            i = Ln * m + k * p + ch
            if kp < 0: F.append(i); f.append(y_past[k * p + ch]); fmap.append(("y", k * p + ch))
            elif spec.tec and kp >= Lh - n: F.append(i); f.append(y_s[ch]); fmap.append(("s", 0))
            else: R.append(i); W.append(qdiag[kp * p + ch]); zs.append(y_s[ch]); widx.append(("Q", kp * p + ch))
    f, W, zs = np.array(f), np.array(W), np.array(zs)
    Wd = None
    if dense:
        # dense weighting matrices (controller.py:708-710): the weight of the free components as one matrix (inputs and outputs do
        # not mix), and a square-root factor of it for the least-squares form (eigendecomposition: W may be singular)
        Wd = np.zeros((len(R), len(R)))
        for a_, (ka, ia) in enumerate(widx):
            for b_, (kb, ib) in enumerate(widx):
                if ka == kb: Wd[a_, b_] = (Rm if ka == "R" else Qm)[ia, ib]
        ev, Ev = np.linalg.eigh(Wd)
        Wh = (Ev * np.sqrt(np.clip(ev, 0.0, None))).T                               # Wh' Wh = Wd
    Qb, _ = np.linalg.qr(M)
    Uf, Sf, Vft = np.linalg.svd(Qb[F], True)
    kf = int(np.sum(Sf > Sf[0] * 1e-9))
    if _parts:
        if dense:
            raise NotImplementedError("solve_nominal_model_based_batch takes diagonal Q, R")
        return dict(Qb=Qb, F=F, R=R, W=W, zs=zs, Uf=Uf, Sf=Sf, Vft=Vft, kf=kf, fmap=fmap, f=f)
    c_p = Vft[:kf].T @ ((Uf[:, :kf].T @ f) / Sf[:kf]); Nn = Vft[kf:].T
    if dense:
        dd = np.linalg.lstsq(Wh @ (Qb[R] @ Nn), Wh @ (zs - Qb[R] @ c_p), rcond=None)[0]
        z = Qb @ (c_p + Nn @ dd)
        dz = z[R] - zs
        return dict(optimal_u=z[:Ln * m][n * m:], cost=float(dz @ Wd @ dz), feas_residual=float(np.max(np.abs(Qb[F] @ c_p - f))))
    sw = np.sqrt(W)
    dd = np.linalg.lstsq(sw[:, None] * (Qb[R] @ Nn), sw * (zs - Qb[R] @ c_p), rcond=None)[0]
    z = Qb @ (c_p + Nn @ dd)
    return dict(optimal_u=z[:Ln * m][n * m:], cost=float(np.sum(W * (z[R] - zs) ** 2)),
                feas_residual=float(np.max(np.abs(Qb[F] @ c_p - f))))
