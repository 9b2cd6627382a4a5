"""Dev tool (CPU only): which a-posteriori quantity separates the instances whose Gram-route solve misses the parity
bars from those that meet them?  Emulates the kernels' route in numpy (G = H H' in fp64, Cholesky, two substitutions)
on the benchmark data and on the seeded random-plant sweep of tests/test_gpu_parity.py, and prints next to the true
errors (against the full-space oracle)

    res   = |t - (H (H' beta) + lam D beta)|_inf / |t|_inf          (exact-Hankel residual)
    dz    = |lam D delta|_inf on the optimal_u rows / |u|_inf,      delta = K^-1 residual (one refinement correction)

    python tools/auto_flag_calib_cpu.py [ncases]
"""
import inspect
import sys

import numpy as np

sys.path.insert(0, "."); sys.path.insert(0, "tests")
import test_gpu_parity as T
from oracle import ddmpc_oracle as orc
from oracle import reduced_form as rf


def one(spec, u_d, y_d, up, yp):
    sol = orc.solve_fullspace(spec, u_d, y_d, up, yp)
    red = rf.solve_reduced(spec, u_d, y_d, up, yp)
    n, m, Ln = spec.n, spec.m, spec.Ln
    H = np.vstack([orc.hankel_matrix(u_d, Ln), orc.hankel_matrix(y_d, Ln)])
    lam = spec.lamb_alpha * spec.eps_max
    beta, t, D, K = red["beta"], red["t"], red["D"], red["K"]
    rho = t - (H @ (H.T @ beta) + lam * D * beta)
    res = np.max(np.abs(rho)) / np.max(np.abs(t))
    delta = np.linalg.solve(K, rho)
    usl = slice(n * m, Ln * m)
    scale = max(np.max(np.abs(sol.optimal_u)), 1e-3)
    dz = np.max(np.abs(lam * D[usl] * delta[usl])) / scale
    dzall = np.max(np.abs(lam * D * delta)) / max(np.max(np.abs(t)), 1e-3)
    dbeta = np.max(np.abs(delta)) / np.max(np.abs(beta))
    eu = np.max(np.abs(red["optimal_u"] - sol.optimal_u)) / scale
    ec = abs(red["cost"] - sol.cost) / max(abs(sol.cost), 1e-6)
    Lc = np.linalg.cholesky(K)
    cond_lb = np.max(np.diag(K)) * np.max(1.0 / np.diag(Lc)) ** 2
    return eu, ec, res, dz, dzall, dbeta, cond_lb


def show(tag, rows):
    a = np.array(rows)
    print("%-44s eu %.1e ec %.1e | res %.1e..%.1e  dz_u %.1e..%.1e  dz_all %.1e..%.1e  dbeta %.1e..%.1e  cond_lb %.1e%s" % (
        tag, a[:, 0].max(), a[:, 1].max(), a[:, 2].min(), a[:, 2].max(), a[:, 3].min(), a[:, 3].max(), a[:, 4].min(),
        a[:, 4].max(), a[:, 5].min(), a[:, 5].max(), a[:, 6].max(),
        "   <-- over" if a[:, 0].max() > 1e-8 or a[:, 1].max() > 1e-9 else ""), flush=True)


for kw in (dict(), dict(tec=False), dict(L=60, N=1000)):
    spec = orc.spec_from_params(**kw)
    N = kw.get("N", 400)
    u_d, y_d, up, yp = T._instances(6, N=N)
    show("four-tank %s" % kw, [one(spec, u_d[b], y_d[b], up[b], yp[b]) for b in range(6)])

src = inspect.getsource(T.test_random_systems_against_oracle)
body = src.split("    B = 3\n")[0].split("(gpu, case, refine):\n", 1)[1]
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 96
for case in range(ncases):
    env = dict(T.__dict__); env["case"] = case
    exec("if True:\n" + body, env)
    spec, plant, N, n = env["spec"], env["plant"], env["N"], env["n"]
    if not spec.robust:
        continue
    diag = np.allclose(spec.Q, np.diag(np.diag(spec.Q))) and np.allclose(spec.R, np.diag(np.diag(spec.R)))
    if not diag or spec.slack != "none":
        continue
    B = 3
    d = T.generate_batch(range(case * 10, case * 10 + B), N=N, plant=plant)
    up = d["u_d"][:, -n:, :].reshape(B, -1).copy(); yp = d["y_d"][:, -n:, :].reshape(B, -1).copy()
    show("case %3d r=%3d" % (case, (spec.m + spec.p) * spec.Ln),
         [one(spec, d["u_d"][b], d["y_d"][b], up[b], yp[b]) for b in range(B)])
