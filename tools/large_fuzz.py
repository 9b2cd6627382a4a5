"""Seeded sweep of problems beyond the register-resident kernels (272 <= (m+p)(L+n) <= ~800 rows) on
ddmpc_large_solve_kernel: random stable plants, m != p, with/without terminal constraint and slack box, scalar and
diagonal weights; every instance against the full-space CPU oracle (status, active-set iterations, optimal_u, cost).

    python tools/large_fuzz.py [--cases 12]
"""
import argparse, sys, time
import numpy as np
sys.path.insert(0, ".")
from direct_data_driven_mpc_amd import _lib as L
from direct_data_driven_mpc_amd.engine import BatchedDDMPC
from direct_data_driven_mpc_amd.harness import generate_batch
from oracle import ddmpc_oracle as orc

ap = argparse.ArgumentParser(); ap.add_argument("--cases", type=int, default=12); a = ap.parse_args()
worst_u = worst_c = 0.0
for case in range(a.cases):
    rng = np.random.default_rng(5000 + case)
    m, p = [(2, 2), (3, 2), (1, 4), (4, 1), (2, 5), (3, 3)][case % 6]
    ns = n = int(rng.integers(2, 5))
    rows = int(rng.integers(280, 780))
    Lh = max(2 * n, rows // (m + p) - n)
    r = (m + p) * (Lh + n)
    N = (m + 1) * (Lh + 2 * n) + int(rng.integers(150, 400))
    eps = 0.002
    slack = "convex" if case % 2 == 1 else "none"
    tec = case % 5 != 4
    dense = case % 4 == 3                                        # dense SPD weighting matrices (round 5: on the phase kernels too)
    if dense:
        def spd(k, sc):
            X = rng.normal(size=(k, k))
            return sc * (np.eye(k) + 0.3 * (X @ X.T) / k)
        Q = spd(p * Lh, 2.0); R = spd(m * Lh, 0.05)
    elif case % 3 == 0:
        Q = 2.0 * np.eye(p * Lh); R = 0.05 * np.eye(m * Lh)
    else:
        Q = np.diag(rng.uniform(1.0, 4.0, p * Lh)); R = np.diag(rng.uniform(0.01, 0.1, m * Lh))
    A = rng.normal(size=(ns, ns)); A *= rng.uniform(0.5, 0.9) / max(abs(np.linalg.eigvals(A)))
    plant = dict(A=A, B=rng.normal(size=(ns, m)), C=rng.normal(size=(p, ns)), D=np.zeros((p, m)), eps_max=eps)
    spec = orc.QPSpec(n=n, m=m, p=p, L=Lh, Q=Q, R=R, u_s=rng.uniform(-0.5, 0.5, m), y_s=rng.uniform(-0.5, 0.5, p),
                      robust=True, eps_max=eps, lamb_alpha=20.0, lamb_sigma=500.0, c=1.0, slack=slack, tec=tec)
    B = 2
    d = generate_batch(range(case * 10, case * 10 + B), N=N, plant=plant)
    up = d["u_d"][:, -n:, :].reshape(B, -1).copy(); yp = d["y_d"][:, -n:, :].reshape(B, -1).copy()
    Qa = Q if (dense or case % 3 == 0) else np.diag(Q); Ra = R if (dense or case % 3 == 0) else np.diag(R)
    with BatchedDDMPC(n=n, m=m, p=p, L_=Lh, N=N, Q=Qa, R=Ra, u_s=spec.u_s, y_s=spec.y_s, batch=B, controller_type=L.ROBUST,
                      slack_type=L.SLACK_CONVEX if slack == "convex" else L.SLACK_NONE, eps_max=eps, lamb_alpha=20.0,
                      lamb_sigma=500.0, c=1.0, use_terminal_constraint=tec) as eng:
        name = eng.kernel_name()
        eng.set_data(d["u_d"], d["y_d"])
        t0 = time.perf_counter(); u, cost, status, iters = eng.solve(up, yp); dt = time.perf_counter() - t0
    eu = ec = 0.0; ok = True
    for b in range(B):
        sol = orc.solve_fullspace(spec, d["u_d"][b], d["y_d"][b], up[b], yp[b])
        ok &= L.STATUS_STRINGS[int(status[b])] == sol.status == "optimal"
        if slack == "convex":
            ok &= int(iters[b]) == sol.iters
        eu = max(eu, np.max(np.abs(u[b] - sol.optimal_u)) / max(np.max(np.abs(sol.optimal_u)), 1e-3))
        ec = max(ec, abs(cost[b] - sol.cost) / max(abs(sol.cost), 1e-6))
    worst_u, worst_c = max(worst_u, eu), max(worst_c, ec)
    print("case %2d m=%d p=%d n=%d L=%3d N=%4d r=%3d slack %-6s tec %d weights %-6s %s: status/iterations %s, iters %s, "
          "rel err u %.1e cost %.1e" % (case, m, p, n, Lh, N, r, slack, tec, "dense" if dense else ("scalar" if case % 3 == 0 else "diag"), name,
                                         "ok" if ok else "MISMATCH", iters.tolist(), eu, ec), flush=True)
    assert ok and "large_solve" in name
print("worst rel err over %d cases: u %.2e, cost %.2e" % (a.cases, worst_u, worst_c))
