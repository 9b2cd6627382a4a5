"""Seeded sweep of the NOMINAL scheme on exact (noise-free) data of random stable plants -- the rank-revealing kernel,
both as the rescue of the register-resident path (<= 271 rows) and as the only kernel beyond it: m, p in 1..4, with the
setpoint a true equilibrium of the plant; every instance against the SVD-based CPU solve (oracle/nominal_exact.py) and the
model-based solution (state-space trajectories).

    python tools/nominal_fuzz.py [--cases 24]
"""
import argparse, sys
import numpy as np
sys.path.insert(0, ".")
from direct_data_driven_mpc_amd import _lib as L
from direct_data_driven_mpc_amd.engine import BatchedDDMPC
from direct_data_driven_mpc_amd.harness import generate_batch
from oracle import ddmpc_oracle as orc
from oracle.nominal_exact import solve_nominal_exact, solve_nominal_model_based

ap = argparse.ArgumentParser(); ap.add_argument("--cases", type=int, default=24)
ap.add_argument("--pipeline", default=None, help="phases | one_workgroup (DDMPC_OPT_LARGE_PIPELINE; default: the library's)")
ap.add_argument("--no-svd", action="store_true", help="compare with the model-based solution only (faster)")
ap.add_argument("--large-only", action="store_true", help="only the cases beyond the register-resident kernels")
a = ap.parse_args()
worst = 0.0; bad = 0; flagged = 0
for case in range(a.cases):
    rng = np.random.default_rng(9000 + case)
    m, p = [(2, 2), (1, 3), (3, 1), (2, 3), (4, 2), (1, 1)][case % 6]
    ns = n = int(rng.integers(2, 5))
    rows = int(rng.integers(60, 260)) if case % 2 == 0 else int(rng.integers(280, 640))
    Lh = max(2 * n, rows // (m + p) - n)
    r = (m + p) * (Lh + n)
    N = (m + 1) * (Lh + 2 * n) + int(rng.integers(100, 300))
    if a.large_only and r <= 271: continue
    A = rng.normal(size=(ns, ns)); A *= rng.uniform(0.5, 0.9) / max(abs(np.linalg.eigvals(A)))
    plant = dict(A=A, B=rng.normal(size=(ns, m)), C=rng.normal(size=(p, ns)), D=np.zeros((p, m)), eps_max=0.0)
    u_s = rng.uniform(-0.5, 0.5, m)
    y_s = (plant["C"] @ np.linalg.inv(np.eye(ns) - A) @ plant["B"]) @ u_s           # a true equilibrium
    q, rw = float(rng.uniform(1.0, 4.0)), float(rng.uniform(0.01, 0.2))
    spec = orc.QPSpec(n=n, m=m, p=p, L=Lh, Q=q * np.eye(p * Lh), R=rw * np.eye(m * Lh), u_s=u_s, y_s=y_s, robust=False,
                      eps_max=0.0, lamb_alpha=0.0, lamb_sigma=0.0, c=0.0, slack="none", tec=True)
    B = 2
    d = generate_batch(range(case * 10, case * 10 + B), N=N, plant=plant)
    up = d["u_d"][:, -n:, :].reshape(B, -1).copy(); yp = d["y_d"][:, -n:, :].reshape(B, -1).copy()
    with BatchedDDMPC(n=n, m=m, p=p, L_=Lh, N=N, Q=q, R=rw, u_s=u_s, y_s=y_s, batch=B, controller_type=L.NOMINAL) as eng:
        name = eng.kernel_name()
        if a.pipeline: eng.set_large_pipeline(a.pipeline)
        eng.set_data(d["u_d"], d["y_d"])
        u, cost, status, _ = eng.solve(up, yp)
    e_svd = e_mod = 0.0; ok = True
    for b in range(B):
        mod = solve_nominal_model_based(spec, plant, up[b], yp[b])
        ref = mod if a.no_svd else solve_nominal_exact(spec, d["u_d"][b], d["y_d"][b], up[b], yp[b])
        # ("optimal_inaccurate": a rank decision without a clear margin is reported, not hidden -- counted, not a failure)
        ok &= (a.no_svd or ref["status"] == "optimal") and L.STATUS_STRINGS[int(status[b])] in ("optimal", "optimal_inaccurate")
        flagged += L.STATUS_STRINGS[int(status[b])] == "optimal_inaccurate"
        sc = max(np.max(np.abs(mod["optimal_u"])), 1e-3)
        e_svd = max(e_svd, np.max(np.abs(u[b] - ref["optimal_u"])) / sc)
        e_mod = max(e_mod, np.max(np.abs(u[b] - mod["optimal_u"])) / sc)
    worst = max(worst, e_mod)
    print("case %2d m=%d p=%d n=%d L=%3d N=%4d r=%3d %-32s status %s  rel err u: vs SVD solve %.1e, vs model-based %.1e" % (
        case, m, p, n, Lh, N, r, name, "ok" if ok else "MISMATCH " + str(status.tolist()), e_svd, e_mod), flush=True)
    bad += (not ok) or not (e_mod < 1e-8)
print("worst rel err vs the model-based solution over %d cases: %.2e; cases off the bars (1e-8) or not optimal: %d; instances reported optimal_inaccurate "
      "(thin rank margin): %d" % (a.cases, worst, bad, flagged))
assert bad == 0
