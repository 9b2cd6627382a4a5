"""Diagnosis of the cases tools/small_fuzz.py reports above tolerance: GPU error next to the error of the numpy reduced form
(oracle/reduced_form.py, same algebra on the CPU) and cond(H) -- equal errors mean the Gram formulation, not the kernel.

    python tools/small_fuzz_diagnose.py [case ...]      (default: the four cases of the 96-case sweep)
"""
import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import test_gpu_parity as T
from direct_data_driven_mpc_amd import _lib as L
from direct_data_driven_mpc_amd.harness import generate_batch
from oracle import ddmpc_oracle as orc
from oracle.reduced_form import solve_reduced
for case in ([int(x) for x in sys.argv[1:]] or [28, 39, 57, 70]):
    rng = np.random.default_rng(1000 + case)
    m, p = [(1, 1), (2, 1), (1, 2), (2, 2), (3, 2), (2, 3)][case % 6]
    ns = int(rng.integers(2, 5)); n = ns
    robust = case % 4 != 3
    Lh = int(rng.integers(2 * n, 2 * n + 9))
    if (m + p) * (Lh + n) > 200: Lh = max(2 * n, 200 // (m + p) - n)
    N = (m + 1) * (Lh + 2 * n) + int(rng.integers(80, 200))
    eps = 0.002
    slack = "convex" if (robust and case % 3 == 1) else "none"
    tec = case % 5 != 4
    wkind = case % 3 if slack == "none" else case % 2
    if wkind == 0: Q = 2.0 * np.eye(p * Lh); R = 0.05 * np.eye(m * Lh)
    elif wkind == 1: Q = np.diag(rng.uniform(1.0, 4.0, p * Lh)); R = np.diag(rng.uniform(0.01, 0.1, m * Lh))
    else: Q = T._spd(rng, p * Lh, 2.0, 2); R = T._spd(rng, m * Lh, 0.05, 2)
    plant = T._random_plant(rng, ns, m, p, eps)
    spec = orc.QPSpec(n=n, m=m, p=p, L=Lh, Q=Q, R=R, u_s=rng.uniform(-0.5, 0.5, m), y_s=rng.uniform(-0.5, 0.5, p),
                      robust=robust, eps_max=eps, lamb_alpha=20.0, lamb_sigma=500.0, c=1.0, slack=slack, tec=tec)
    B = 3
    d = generate_batch(range(case * 10, case * 10 + B), N=N, plant=plant)
    up = d["u_d"][:, -n:, :].reshape(B, -1).copy(); yp = d["y_d"][:, -n:, :].reshape(B, -1).copy()
    with T._engine(spec, N, B) as eng:
        name = eng.kernel_name()
        eng.set_data(d["u_d"], d["y_d"])
        u, cost, status, iters = eng.solve(up, yp)
    for b in range(B):
        sol = orc.solve_fullspace(spec, d["u_d"][b], d["y_d"][b], up[b], yp[b])
        H = np.vstack([orc.hankel_matrix(d["u_d"][b], Lh + n), orc.hankel_matrix(d["y_d"][b], Lh + n)])
        sv = np.linalg.svd(H, compute_uv=False)
        scale = max(np.max(np.abs(sol.optimal_u)), 1e-3)
        eu = np.max(np.abs(u[b] - sol.optimal_u)) / scale; ec = abs(cost[b] - sol.cost) / max(abs(sol.cost), 1e-6)
        red = ""
        if wkind != 2:
            try:
                rd = solve_reduced(spec, d["u_d"][b], d["y_d"][b], up[b], yp[b])
                ur = rd["optimal_u"] if isinstance(rd, dict) else rd.optimal_u
                cr = rd["cost"] if isinstance(rd, dict) else rd.cost
                red = " | numpy reduced form: u %.1e cost %.1e" % (np.max(np.abs(np.ravel(ur) - sol.optimal_u)) / scale, abs(cr - sol.cost) / max(abs(sol.cost), 1e-6))
            except Exception as e:
                red = " | reduced: " + str(e)[:40]
        print("case %d b %d %s m=%d p=%d n=%d L=%d N=%d robust=%d slack=%s tec=%d w=%d status %d/%s: u %.2e cost %.2e cond(H) %.1e cost=%.3e%s" % (
            case, b, name, m, p, n, Lh, N, robust, slack, tec, wkind, status[b], sol.status, eu, ec, sv[0] / sv[-1], sol.cost, red), flush=True)
