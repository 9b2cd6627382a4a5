"""Second, independent CPU formulation: the reduced r x r system the HIP kernels
solve, written in plain numpy.  TEST INFRASTRUCTURE ONLY (never imported by the
product).  It exists so that (a) the algebra behind the kernels is checked
against the full-space oracle (`ddmpc_oracle.solve_fullspace`) on the CPU, and
(b) kernel intermediates (Gram, K, beta) can be compared when debugging.

Derivation (robust scheme, direct_data_driven_mpc_controller.py:506-547,679-722)
-----------------------------------------------------------------------------
With z = [ubar; ybar + sigma] = H alpha (H = [Hu; Hy], r rows) the optimal
alpha is the minimum-norm one, alpha = H' beta, G = H H', z = G beta and
||alpha||^2 = beta' G beta.  Eliminating sigma (it only appears in diagonal
quadratics when Q is diagonal) leaves, per component i of z, either a hard
value (ubar on the internal/terminal windows) or a quadratic penalty
w_i (z_i - t_i)^2.  Stationarity then reads

    (G + lam * D) beta = t,     D = diag(1 / w_i)  (0 for hard components),

with lam = lamb_alpha * eps_max, an SPD r x r system.  z = t - lam*D*beta.
The slack box |sigma_k| <= c*eps_max, k in the prediction window (:659,674),
switches (w_i, t_i) of the affected component; the primal-dual active-set rule
is "sigma_hat_i = -lam*beta_i/lamb_sigma outside the box <=> bound active".
The nominal scheme is lam = 0 with sigma absent (G beta = t, exact whenever the
noisy Hankel matrix has full row rank).
"""
from __future__ import annotations

import numpy as np

from .ddmpc_oracle import QPSpec, hankel_matrix


def component_tables(spec: QPSpec, u_past, y_past, act=None):
    """Per-component (D_ii, t_i) in the natural stacking z=[ubar(Ln*m); w(Ln*p)].

    act: signed active set over the L*p boxed slack components (k=0..L-1).
    """
    n, m, p, L, Ln = spec.n, spec.m, spec.p, spec.L, spec.Ln
    u_s = np.asarray(spec.u_s, float).reshape(-1)
    y_s = np.asarray(spec.y_s, float).reshape(-1)
    u_past = np.asarray(u_past, float).reshape(-1)
    y_past = np.asarray(y_past, float).reshape(-1)
    rdiag = np.diag(spec.R)
    qdiag = np.diag(spec.Q)
    if not (np.allclose(spec.R, np.diag(rdiag)) and np.allclose(spec.Q, np.diag(qdiag))):
        raise NotImplementedError("reduced form assumes diagonal Q, R")
    r = Ln * (m + p)
    D = np.zeros(r)
    t = np.zeros(r)
    bound = spec.c * spec.eps_max if (spec.robust and spec.slack == "convex") else np.inf
    if act is None:
        act = np.zeros(L * p, dtype=int)
    for k in range(Ln):          # k = 0..Ln-1  <->  time index k-n
        kp = k - n               # prediction index, <0 on the internal window
        is_int = kp < 0
        is_term = spec.tec and kp >= L - n
        for ch in range(m):
            i = k * m + ch
            if is_int:
                D[i], t[i] = 0.0, u_past[k * m + ch]
            elif is_term:
                D[i], t[i] = 0.0, u_s[ch]
            else:
                D[i], t[i] = 1.0 / rdiag[kp * m + ch], u_s[ch]
        for ch in range(p):
            i = Ln * m + k * p + ch
            if not spec.robust:
                if is_int:
                    D[i], t[i] = 0.0, y_past[k * p + ch]
                elif is_term:
                    D[i], t[i] = 0.0, y_s[ch]
                else:
                    D[i], t[i] = 1.0 / qdiag[kp * p + ch], y_s[ch]
                continue
            ls = spec.lamb_sigma
            if is_int:
                D[i], t[i] = 1.0 / ls, y_past[k * p + ch]
                continue
            s = act[kp * p + ch]
            if is_term:
                if s == 0:
                    D[i], t[i] = 1.0 / ls, y_s[ch]
                else:
                    D[i], t[i] = 0.0, y_s[ch] + s * bound
            else:
                q = qdiag[kp * p + ch]
                if s == 0:
                    D[i], t[i] = 1.0 / q + 1.0 / ls, y_s[ch]
                else:
                    D[i], t[i] = 1.0 / q, y_s[ch] + s * bound
    return D, t


def solve_reduced(spec: QPSpec, u_d, y_d, u_past, y_past, max_iter=50):
    n, m, p, L, Ln = spec.n, spec.m, spec.p, spec.L, spec.Ln
    H = np.vstack([hankel_matrix(u_d, Ln), hankel_matrix(y_d, Ln)])
    G = H @ H.T
    lam = spec.lamb_alpha * spec.eps_max if spec.robust else 0.0
    boxed = spec.robust and spec.slack == "convex"
    bound = spec.c * spec.eps_max if boxed else np.inf
    act = np.zeros(L * p, dtype=int)
    iters = 0
    status = "optimal"
    while True:
        iters += 1
        D, t = component_tables(spec, u_past, y_past, act)
        K = G + lam * np.diag(D)
        Lc = np.linalg.cholesky(K)
        beta = np.linalg.solve(Lc.T, np.linalg.solve(Lc, t))
        if not boxed:
            break
        w_pred = slice(Ln * m + n * p, Ln * (m + p))
        sig_hat = -lam * beta[w_pred] / spec.lamb_sigma
        new = np.where(sig_hat > bound, 1, np.where(sig_hat < -bound, -1, 0))
        if np.array_equal(new, act):
            break
        act = new
        if iters >= max_iter:
            status = "solver_error"
            break
    z = t - lam * D * beta
    ubar = z[:Ln * m]
    w = z[Ln * m:]
    u_s = np.tile(np.asarray(spec.u_s, float), L)
    y_s = np.tile(np.asarray(spec.y_s, float), L)
    if spec.robust:
        sigma = np.zeros(Ln * p)
        yfix = np.asarray(y_past, float).reshape(-1)
        sigma[:n * p] = w[:n * p] - yfix
        sh = -lam * beta[Ln * m + n * p:] / spec.lamb_sigma
        sg = np.where(act != 0, act * bound, sh) if boxed else sh
        sigma[n * p:] = sg
        if spec.tec:
            sigma[L * p:] = w[L * p:] - np.tile(np.asarray(spec.y_s, float), n)
        ybar = w - sigma
    else:
        sigma = None
        ybar = w
    alpha = H.T @ beta
    du = ubar[n * m:] - u_s
    dy = ybar[n * p:] - y_s
    cost = float(du @ (np.diag(spec.R) * du) + dy @ (np.diag(spec.Q) * dy))
    if spec.robust:
        cost += float(lam * beta @ (G @ beta) + spec.lamb_sigma * sigma @ sigma)
    return dict(status=status, optimal_u=ubar[n * m:].copy(), cost=cost, beta=beta,
                alpha=alpha, ubar=ubar, ybar=ybar, sigma=sigma, iters=iters, act=act,
                G=G, K=K, t=t, D=D)
