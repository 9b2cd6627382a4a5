"""BASELINE configs[4] (nominal, m=p=8, n=8, L=30, N=2000, B=512, exact data of a random stable plant) on the
rank-revealing kernel: accuracy against the SVD-based CPU solve on the first `--check` instances (oracle solves in
forked workers, before the GPU runtime starts), and time per batch.

    python tools/config5_check.py [--check 64]
"""
import argparse, sys, time
import multiprocessing as mp
import numpy as np
sys.path.insert(0, ".")
from direct_data_driven_mpc_amd.harness import generate_batch
from oracle import ddmpc_oracle as orc
from oracle.nominal_exact import solve_nominal_exact

ap = argparse.ArgumentParser(); ap.add_argument("--check", type=int, default=64)
ap.add_argument("--robust", action="store_true", help="the ROBUST scheme (slack CONVEX) on noisy data of the same plant: "
                "ddmpc_large_solve_kernel against the full-space oracle")
a = ap.parse_args()
rng = np.random.default_rng(0)
ns = n = 8; m = p = 8; Lh = 30; N = 2000; B = 512
A = rng.normal(size=(ns, ns)); A *= 0.9 / max(abs(np.linalg.eigvals(A)))
plant = dict(A=A, B=rng.normal(size=(ns, m)), C=rng.normal(size=(p, ns)), D=np.zeros((p, m)), eps_max=0.0)
u_s = 0.1 * np.ones(m); y_s = (plant["C"] @ np.linalg.inv(np.eye(ns) - A) @ plant["B"]) @ u_s
spec = orc.QPSpec(n=n, m=m, p=p, L=Lh, Q=3.0 * np.eye(p * Lh), R=1e-4 * np.eye(m * Lh), u_s=u_s, y_s=y_s, robust=False,
                  eps_max=0.0, lamb_alpha=0.0, lamb_sigma=0.0, c=0.0, slack="none", tec=True)
if a.robust:
    plant["eps_max"] = 0.002
    spec = orc.QPSpec(n=n, m=m, p=p, L=Lh, Q=3.0 * np.eye(p * Lh), R=1e-4 * np.eye(m * Lh), u_s=u_s, y_s=y_s, robust=True,
                      eps_max=0.002, lamb_alpha=50.0, lamb_sigma=1000.0, c=1.0, slack="convex", tec=True)
d = generate_batch(range(B), N=N, plant=plant)
up = d["u_d"][:, -n:, :].reshape(B, -1).copy(); yp = d["y_d"][:, -n:, :].reshape(B, -1).copy()


def _ref(b):
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(limits=1)
    except Exception:
        pass
    if a.robust:
        r = orc.solve_fullspace(spec, d["u_d"][b], d["y_d"][b], up[b], yp[b])
        return r.status, r.optimal_u, r.cost, r.iters
    r = solve_nominal_exact(spec, d["u_d"][b], d["y_d"][b], up[b], yp[b])
    return r["status"], r["optimal_u"], r["cost"], r["rank"]


K = min(a.check, B)
t0 = time.perf_counter()
with mp.get_context("fork").Pool(16) as pool:
    refs = pool.map(_ref, range(K))
print("oracle (%s): %d instances in %.1f s on 16 worker processes" % ("full-space KKT" if a.robust else "SVD route", K, time.perf_counter() - t0), flush=True)

from direct_data_driven_mpc_amd import _lib as L
from direct_data_driven_mpc_amd.engine import BatchedDDMPC
if a.robust:
    eng = BatchedDDMPC(n=n, m=m, p=p, L_=Lh, N=N, Q=3.0, R=1e-4, u_s=u_s, y_s=y_s, batch=B, controller_type=L.ROBUST,
                       slack_type=L.SLACK_CONVEX, eps_max=0.002, lamb_alpha=50.0, lamb_sigma=1000.0, c=1.0)
else:
    eng = BatchedDDMPC(n=n, m=m, p=p, L_=Lh, N=N, Q=3.0, R=1e-4, u_s=u_s, y_s=y_s, batch=B, controller_type=L.NOMINAL)
eng.set_data(d["u_d"], d["y_d"])
u, cost, status, _ = eng.solve(up, yp)
t = time.perf_counter(); u, cost, status, its = eng.solve(up, yp); dt = time.perf_counter() - t
if a.robust:
    print("cfg5 size, robust scheme with slack box, kernel %s, B=%d: %.1f ms per batch -> %.3e solves/s; status values %s, "
          "active-set iterations %d..%d" % (eng.kernel_name(), B, dt * 1e3, B / dt, sorted(set(status.tolist())), its.min(), its.max()))
    eu = max(np.max(np.abs(u[b] - refs[b][1])) / np.max(np.abs(refs[b][1])) for b in range(K))
    ec = max(abs(cost[b] - refs[b][2]) / abs(refs[b][2]) for b in range(K))
    print("first %d instances vs the full-space oracle (all '%s'): max rel err u %.3e, cost %.3e; iterations equal: %s" % (
        K, ",".join(sorted(set(r[0] for r in refs))), eu, ec, all(int(its[b]) == refs[b][3] for b in range(K))))
    sys.exit(0)
print("cfg5 nominal exact, kernel %s, B=%d: %.1f ms per batch -> %.3e solves/s; status values %s" % (
    eng.kernel_name(), B, dt * 1e3, B / dt, sorted(set(status.tolist()))))
eu = max(np.max(np.abs(u[b] - refs[b][1])) / np.max(np.abs(refs[b][1])) for b in range(K))
ec = max(abs(cost[b] - refs[b][2]) / abs(refs[b][2]) for b in range(K))
print("first %d instances vs the SVD-based oracle (all '%s', rank %d of 608): max rel err u %.3e, cost %.3e" % (
    K, ",".join(sorted(set(r[0] for r in refs))), refs[0][3], eu, ec))


def model_based_solution(b):
    """The same constrained least-squares problem on a basis of the plant's trajectory space built from (A, B, C) --
    not data-driven, well conditioned: a yardstick for GPU and oracle alike (exact data only)."""
    Ln = Lh + n
    A_, B_, C_ = plant["A"], plant["B"], plant["C"]
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
    F, R, f, W, zs = [], [], [], [], []
    for k in range(Ln):
        kp = k - n
        for ch in range(m):
            i = k * m + ch
            if kp < 0: F.append(i); f.append(up[b][k * m + ch])
            elif kp >= Lh - n: F.append(i); f.append(u_s[ch])
            else: R.append(i); W.append(1e-4); zs.append(u_s[ch])
    for k in range(Ln):
        kp = k - n
        for ch in range(p):
            i = Ln * m + k * p + ch
            if kp < 0: F.append(i); f.append(yp[b][k * p + ch])
            elif kp >= Lh - n: F.append(i); f.append(y_s[ch])
            else: R.append(i); W.append(3.0); zs.append(y_s[ch])
    f, W, zs = np.array(f), np.array(W), np.array(zs)
    Qb, _ = np.linalg.qr(M)
    Uf, Sf, Vft = np.linalg.svd(Qb[F], True)
    kf = int(np.sum(Sf > Sf[0] * 1e-9))
    c_p = Vft[:kf].T @ ((Uf[:, :kf].T @ f) / Sf[:kf]); Nn = Vft[kf:].T
    sw = np.sqrt(W)
    dd = np.linalg.lstsq(sw[:, None] * (Qb[R] @ Nn), sw * (zs - Qb[R] @ c_p), rcond=None)[0]
    z = Qb @ (c_p + Nn @ dd)
    return z[:Ln * m][n * m:]


errs = np.array([np.max(np.abs(u[b] - refs[b][1])) / np.max(np.abs(refs[b][1])) for b in range(K)])
print("instances above 5e-9 against the oracle: %d of %d" % (int(np.sum(errs > 5e-9)), K))
for b in np.argsort(-errs)[:4]:
    ut = model_based_solution(b); sc = np.max(np.abs(ut))
    print("instance %3d: GPU vs oracle %.2e | GPU vs model-based solution %.2e | oracle vs model-based solution %.2e" % (
        b, errs[b], np.max(np.abs(u[b] - ut)) / sc, np.max(np.abs(refs[b][1] - ut)) / sc))
