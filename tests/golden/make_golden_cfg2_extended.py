"""Golden solutions of BASELINE configs[1] (four-tank robust DD-MPC, L = 30, N = 400) and configs[3] (L = 60, N = 1000), slack NONE
and CONVEX, in EXTENDED precision:
the QP exactly as the reference hands it to CVXPY -- variables [alpha; ubar; ybar; sigma], the cost of
direct_data_driven_mpc_controller.py:679-722, the equalities of :506-629, the slack box of :631-677 -- assembled by
oracle/ddmpc_oracle.build_fullspace_qp (every entry of P, q, A, b is a data value or a weight: exact in 80-bit) and solved through
its dense KKT system by Gaussian elimination with partial pivoting and iterative refinement, all in np.longdouble (eps 1.1e-19).
Slack CONVEX: the active set of the fp64 primal-dual iteration is taken as the candidate and its optimality conditions (bounds on
the inactive sigma, multiplier signs on the active ones) are verified IN extended precision before the result is stored.

    python tests/golden/make_golden_cfg2_extended.py          # ~5 minutes; writes tests/golden/cfg2_extended.npz, cfg4_extended.npz

TEST INFRASTRUCTURE ONLY.  What it pins: the accuracy of the fp64 checkers (numpy full-space oracle, compiled C restatement) and of
the GPU on the headline configuration against a solution three orders of magnitude beyond fp64 -- not the reference's own solver
output (CVXPY is not installed anywhere in this pipeline; its OSQP stops at ~1e-5 anyway)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from direct_data_driven_mpc_amd.harness import generate_batch          # noqa: E402  (bit-exact with the reference's data generation)
from oracle import ddmpc_oracle as orc                                 # noqa: E402

LD = np.longdouble


def lu_solve_ld(K, rhs, refine=3):
    """K x = rhs in long double: LU with partial pivoting (row operations vectorised), then `refine` steps of iterative refinement."""
    n = K.shape[0]
    Aw = K.copy()
    piv = np.arange(n)
    for k in range(n - 1):
        p = k + int(np.argmax(np.abs(Aw[k:, k])))
        if p != k:
            Aw[[k, p]] = Aw[[p, k]]
            piv[[k, p]] = piv[[p, k]]
        f = Aw[k + 1:, k] / Aw[k, k]
        Aw[k + 1:, k] = f
        Aw[k + 1:, k + 1:] -= np.outer(f, Aw[k, k + 1:])

    def solve(b):
        y = b[piv].copy()
        for k in range(n):                      # L y = P b (unit lower)
            y[k + 1:] -= Aw[k + 1:, k] * y[k]
        for k in range(n - 1, -1, -1):          # U x = y
            y[k] /= Aw[k, k]
            y[:k] -= Aw[:k, k] * y[k]
        return y
    x = solve(rhs)
    for _ in range(refine):
        x = x + solve(rhs - K @ x)
    return x


def solve_extended(spec, u_d, y_d, u_past, y_past):
    qp = orc.build_fullspace_qp(spec, u_d, y_d, u_past, y_past)
    ref = orc.solve_fullspace(spec, u_d, y_d, u_past, y_past)           # fp64: supplies the candidate active set only
    act = ref.active
    idx = np.nonzero(act)[0]
    nx = qp.P.shape[0]
    Eb = np.zeros((idx.size, nx)); Eb[np.arange(idx.size), qp.box_idx[idx]] = 1.0
    A2 = np.vstack([qp.A, Eb]).astype(LD)
    b2 = np.concatenate([qp.b, act[idx] * qp.bound]).astype(LD)
    ne = A2.shape[0]
    K = np.zeros((nx + ne, nx + ne), dtype=LD)
    K[:nx, :nx] = 2 * qp.P.astype(LD); K[:nx, nx:] = A2.T; K[nx:, :nx] = A2
    rhs = np.concatenate([-qp.q.astype(LD), b2])
    sol = lu_solve_ld(K, rhs)
    x, nu = sol[:nx], sol[nx:]
    res = float(np.max(np.abs(K @ sol - rhs)) / np.max(np.abs(rhs)))
    # optimality of the active set, in extended precision
    if qp.box_idx.size:
        sig = x[qp.box_idx]
        mu = np.zeros(qp.box_idx.size, dtype=LD); mu[idx] = nu[qp.A.shape[0]:]
        assert np.all(np.abs(sig[act == 0]) <= qp.bound), "an inactive sigma violates its bound"
        assert np.all(mu[act == 1] > 0) and np.all(mu[act == -1] < 0), "a multiplier of an active bound has the wrong sign"
    cost = x @ (qp.P.astype(LD) @ x) + qp.q.astype(LD) @ x + LD(qp.const)
    sl = qp.sl
    ubar = x[sl["ubar"]]
    return dict(optimal_u=np.asarray(ubar[spec.n * spec.m:], dtype=np.float64), cost=float(cost), kkt_residual=res,
                alpha=np.asarray(x[sl["alpha"]], dtype=np.float64), iters=int(ref.iters), n_active=int(idx.size),
                fp64_u=ref.optimal_u, fp64_cost=ref.cost)


def make(fname, seeds, L_, N):
    d = generate_batch(seeds, N=N)
    out = dict(seeds=np.array(seeds), L=np.array(L_), N=np.array(N))
    for tag, kw in (("none", dict()), ("convex", dict(slack_var_constraint_type=1))):
        spec = orc.spec_from_params(L=L_, N=N, **kw)
        n = spec.n
        U, C, AL, IT, NA = [], [], [], [], []
        for k, s in enumerate(seeds):
            up = d["u_d"][k, -n:, :].reshape(-1); yp = d["y_d"][k, -n:, :].reshape(-1)
            r = solve_extended(spec, d["u_d"][k], d["y_d"][k], up, yp)
            eu = np.max(np.abs(r["fp64_u"] - r["optimal_u"])) / np.max(np.abs(r["optimal_u"]))
            print("%s slack %-6s seed %4d: KKT residual %.1e, active bounds %2d (fp64 iterations %d); fp64 full-space oracle vs extended: u %.1e cost %.1e" % (
                fname, tag, s, r["kkt_residual"], r["n_active"], r["iters"], eu, abs(r["fp64_cost"] - r["cost"]) / abs(r["cost"])), flush=True)
            U.append(r["optimal_u"]); C.append(r["cost"]); AL.append(r["alpha"]); IT.append(r["iters"]); NA.append(r["n_active"])
        out["optimal_u_" + tag] = np.array(U); out["cost_" + tag] = np.array(C); out["alpha_" + tag] = np.array(AL)
        out["iters_" + tag] = np.array(IT); out["n_active_" + tag] = np.array(NA)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), fname), **out)
    print("wrote", fname)


if __name__ == "__main__":
    make("cfg2_extended.npz", [0, 1, 2, 3, 1000, 4095], 30, 400)          # BASELINE configs[1]
    make("cfg4_extended.npz", [0, 1023], 60, 1000)                          # BASELINE configs[3]: 1,321 variables + 392 equalities
