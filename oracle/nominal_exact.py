"""TEST INFRASTRUCTURE ONLY -- CPU check for the nominal scheme on rank-deficient (noise-free) data.

The nominal QP of the reference (direct_data_driven_mpc_controller.py:506-538,549-629,679-711):

    min  (ubar_P - 1 x u_s)' R (...) + (ybar_P - 1 x y_s)' Q (...)
    s.t. [ubar; ybar] = H alpha,  internal window = past data,  [TEC] terminal window = setpoint

has a unique [ubar; ybar] but no unique alpha when H is rank-deficient.  This solver works on H itself with
orthogonal factorisations only (SVD of H for a basis of its range, SVD of the constraint block for its null
space, least squares for the rest) -- deliberately a different route from the Gram/Cholesky one the GPU
kernels take.  Parity unpinned against the reference (cvxpy absent), like the rest of the QP oracle.
Diagonal Q, R only."""
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
