"""Golden solutions of BASELINE configs[4] (nominal scheme, m = p = 8, n = 8, L = 30, N = 2000, exact data) in EXTENDED precision,
from the DATA alone -- the reference's formulation (direct_data_driven_mpc_controller.py:506-538,549-629,679-711):

    min (ubar_P - 1 x u_s)' R (...) + (ybar_P - 1 x y_s)' Q (...)   s.t.  [ubar; ybar] = H alpha,  internal window = past data,
                                                                          terminal window = setpoint

solved with orthogonal factorisations only (no Gram matrix), all in 80-bit long double (eps 1.1e-19): column-pivoted Householder
QR of H' for a basis of range(H), column-pivoted QR of the constraint block for its null space, QR least squares for the rest.
cond(H) ~ 1e4..1e6 on these data, so the result is good to ~1e-13 -- three orders beyond what the fp64 SVD route of
oracle/nominal_exact.py reaches (one instance of the batch, 283, sits at 1e-8 there, ON the parity bar), and it never sees the
plant's (A, B, C), which the reference controller does not have either.

    python tests/golden/make_golden_cfg5.py           # ~1 minute per instance; writes tests/golden/cfg5_extended.npz

TEST INFRASTRUCTURE ONLY.  Sixteen instances of the batch, among them 283 (the one the fp64 oracles disagree on) and 256, 491,
311 (the next largest distances in profiles/r03e_cfg5_parity.log)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from direct_data_driven_mpc_amd.harness import generate_batch          # noqa: E402  (bit-exact with the reference's data generation)
from oracle import ddmpc_oracle as orc                                 # noqa: E402

LD = np.longdouble


def qr_pivoted(A, tol):
    """Householder QR with column pivoting of A (rows x cols, long double), stopped at the numerical rank (largest remaining
    column norm <= tol * the first).  Returns (V, betas, R, perm, k): reflectors, R (k x cols, upper trapezoidal in the
    pivoted order), the column order, the rank."""
    A = A.copy()
    rows, cols = A.shape
    perm = np.arange(cols)
    norms = np.sum(A * A, axis=0)
    V, betas = [], []
    first = None
    k = 0
    for j in range(min(rows, cols)):
        norms[j:] = np.sum(A[j:, j:] * A[j:, j:], axis=0)                 # (recomputed: no down-dating error)
        pj = j + int(np.argmax(norms[j:]))
        if first is None:
            first = norms[pj]
        if norms[pj] <= tol * tol * first:
            break
        if pj != j:
            A[:, [j, pj]] = A[:, [pj, j]]; perm[[j, pj]] = perm[[pj, j]]
        x = A[j:, j].copy()
        alpha = -np.copysign(np.sqrt(np.sum(x * x)), x[0])
        v = x; v[0] -= alpha
        beta = LD(2) / np.sum(v * v)
        A[j:, j:] -= np.outer(v, beta * (v @ A[j:, j:]))
        V.append(v); betas.append(beta)
        k += 1
    return V, betas, np.triu(A[:k, :]), perm, k


def apply_qt(V, betas, b):
    """Q' b for the reflectors of qr_pivoted."""
    b = b.copy()
    for j, (v, beta) in enumerate(zip(V, betas)):
        b[j:] -= v * (beta * (v @ b[j:]))
    return b


def apply_q(V, betas, b):
    b = b.copy()
    for j in range(len(V) - 1, -1, -1):
        v, beta = V[j], betas[j]
        b[j:] -= v * (beta * (v @ b[j:]))
    return b


def back_substitute(R, y):
    n = R.shape[0]
    x = np.zeros(n, dtype=LD)
    for i in range(n - 1, -1, -1):
        x[i] = (y[i] - R[i, i + 1:] @ x[i + 1:]) / R[i, i]
    return x


def forward_substitute_t(R, y):          # R' x = y
    n = R.shape[0]
    x = np.zeros(n, dtype=LD)
    for i in range(n):
        x[i] = (y[i] - R[:i, i] @ x[:i]) / R[i, i]
    return x


def solve_extended(spec, u_d, y_d, u_past, y_past):
    n, m, p, L_ = spec.n, spec.m, spec.p, spec.L
    Ln = L_ + n
    H = np.vstack([orc.hankel_matrix(u_d, Ln), orc.hankel_matrix(y_d, Ln)]).astype(LD)       # rows: ubar (k m + ch), then ybar
    fixed, weight, target = {}, {}, {}
    for k in range(Ln):
        kp = k - n
        for ch in range(m):
            i = k * m + ch
            if kp < 0: fixed[i] = u_past[k * m + ch]                                          # controller.py:577-581
            elif spec.tec and kp >= L_ - n: fixed[i] = spec.u_s[ch]                            # :612-627
            else: weight[i] = spec.R[kp * m + ch, kp * m + ch]; target[i] = spec.u_s[ch]       # :708-710
        for ch in range(p):
            i = Ln * m + k * p + ch
            if kp < 0: fixed[i] = y_past[k * p + ch]
            elif spec.tec and kp >= L_ - n: fixed[i] = spec.y_s[ch]
            else: weight[i] = spec.Q[kp * p + ch, kp * p + ch]; target[i] = spec.y_s[ch]
    F, Rr = sorted(fixed), sorted(weight)
    f = np.array([fixed[i] for i in F], dtype=LD); W = np.array([weight[i] for i in Rr], dtype=LD)
    zs = np.array([target[i] for i in Rr], dtype=LD)
    # basis of range(H): H' P = Q R  =>  H = P R' Q'  =>  range(H) = range(M), M = (R')[inverse row order]   (r x k)
    V, be, R, perm, k = qr_pivoted(H.T.copy(), LD(1e-13))
    M = np.zeros((H.shape[0], k), dtype=LD)
    M[perm, :] = R.T
    # constraints M_F v = f: M_F' Pc = Qc Rc (k x nF, rank kf)  =>  v = Qc [y1; y2], Rc1' y1 = (Pc' f)[:kf]; consistency of the rest
    Vc, bc, Rc, pc, kf = qr_pivoted(M[F].T.copy(), LD(1e-13))
    fp = f[pc]
    y1 = forward_substitute_t(Rc[:, :kf], fp[:kf])
    feas = float(np.max(np.abs(Rc[:, kf:].T @ y1 - fp[kf:]))) if kf < len(F) else 0.0
    # free part: min | sqrt(W) (M_R Qc [y1; y2] - zs) |  over y2
    MQ = np.array([apply_qt(Vc, bc, M[i].copy()) for i in Rr])                                 # rows of M_R Qc
    sw = np.sqrt(W)
    A2 = sw[:, None] * MQ[:, kf:]
    rhs = sw * (zs - MQ[:, :kf] @ y1)
    V2, b2, R2, p2, k2 = qr_pivoted(A2, LD(1e-13))
    assert k2 == A2.shape[1], "the cost pins every remaining direction"
    qtr = apply_qt(V2, b2, rhs)
    y2p = back_substitute(R2[:, :k2], qtr[:k2])
    y2 = np.zeros(k2, dtype=LD); y2[p2] = y2p
    v = apply_q(Vc, bc, np.concatenate([y1, y2]))
    z = M @ v
    cost = np.sum(W * (z[Rr] - zs) ** 2)
    return dict(optimal_u=np.asarray(z[:Ln * m][n * m:], dtype=np.float64), cost=float(cost), rank=int(k), rank_fixed=int(kf),
                feas_residual=feas, z=np.asarray(z, dtype=np.float64))


def config5(B):
    rng = np.random.default_rng(0)
    ns = n = 8; m = p = 8; Lh = 30; N = 2000
    A = rng.normal(size=(ns, ns)); A *= 0.9 / max(abs(np.linalg.eigvals(A)))
    plant = dict(A=A, B=rng.normal(size=(ns, m)), C=rng.normal(size=(p, ns)), D=np.zeros((p, m)), eps_max=0.0)
    u_s = 0.1 * np.ones(m); y_s = (plant["C"] @ np.linalg.inv(np.eye(ns) - A) @ plant["B"]) @ u_s
    spec = orc.QPSpec(n=n, m=m, p=p, L=Lh, Q=3.0 * np.eye(p * Lh), R=1e-4 * np.eye(m * Lh), u_s=u_s, y_s=y_s, robust=False,
                      eps_max=0.0, lamb_alpha=0.0, lamb_sigma=0.0, c=0.0, slack="none", tec=True)
    return spec, plant, N


if __name__ == "__main__":
    assert np.finfo(LD).eps < 2e-19, "needs 80-bit long double"
    inst = [283, 0, 256, 491, 311, 1, 2, 3, 50, 100, 150, 200, 300, 400, 500, 511]
    spec, plant, N = config5(512)
    n = spec.n
    out = dict(instances=np.array(inst))
    us, cs = [], []
    for b in inst:
        d = generate_batch([b], N=N, plant=plant)
        up = d["u_d"][0, -n:, :].reshape(-1); yp = d["y_d"][0, -n:, :].reshape(-1)
        t0 = time.time()
        sol = solve_extended(spec, d["u_d"][0], d["y_d"][0], up, yp)
        print("instance %3d: rank %d (fixed block %d), constraint residual %.1e, cost %.15e  [%.0f s]" % (
            b, sol["rank"], sol["rank_fixed"], sol["feas_residual"], sol["cost"], time.time() - t0), flush=True)
        us.append(sol["optimal_u"]); cs.append(sol["cost"])
    out["optimal_u"] = np.array(us); out["cost"] = np.array(cs)
    np.savez(os.path.join(os.path.dirname(os.path.abspath(__file__)), "cfg5_extended.npz"), **out)
    print("wrote cfg5_extended.npz")
