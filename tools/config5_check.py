"""Dev tool: BASELINE configs[4] (nominal, m=p=8, n=8, L=30, N=2000, B=512, exact data) on the rank-revealing kernel:
accuracy against the SVD-based CPU solve on a few instances, and time per batch."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from direct_data_driven_mpc_amd import _lib as L
from direct_data_driven_mpc_amd.engine import BatchedDDMPC
from direct_data_driven_mpc_amd.harness import generate_batch
from oracle import ddmpc_oracle as orc
from oracle.nominal_exact import solve_nominal_exact

rng = np.random.default_rng(0)
ns = n = 8; m = p = 8; Lh = 30; N = 2000; B = 512
A = rng.normal(size=(ns, ns)); A *= 0.9 / max(abs(np.linalg.eigvals(A)))
plant = dict(A=A, B=rng.normal(size=(ns, m)), C=rng.normal(size=(p, ns)), D=np.zeros((p, m)), eps_max=0.0)
u_s = 0.1 * np.ones(m); y_s = (plant["C"] @ np.linalg.inv(np.eye(ns) - A) @ plant["B"]) @ u_s
spec = orc.QPSpec(n=n, m=m, p=p, L=Lh, Q=3.0 * np.eye(p * Lh), R=1e-4 * np.eye(m * Lh), u_s=u_s, y_s=y_s, robust=False,
                  eps_max=0.0, lamb_alpha=0.0, lamb_sigma=0.0, c=0.0, slack="none", tec=True)
d = generate_batch(range(B), N=N, plant=plant)
up = d["u_d"][:, -n:, :].reshape(B, -1).copy(); yp = d["y_d"][:, -n:, :].reshape(B, -1).copy()
eng = BatchedDDMPC(n=n, m=m, p=p, L_=Lh, N=N, Q=3.0, R=1e-4, u_s=u_s, y_s=y_s, batch=B, controller_type=L.NOMINAL)
eng.set_data(d["u_d"], d["y_d"])
u, cost, status, _ = eng.solve(up, yp)
t = time.perf_counter(); u, cost, status, _ = eng.solve(up, yp); dt = time.perf_counter() - t
print("cfg5 nominal exact, B=%d: %.1f ms per batch -> %.3e solves/s; status %s" % (B, dt * 1e3, B / dt, sorted(set(status.tolist()))))
for b in range(4):
    ref = solve_nominal_exact(spec, d["u_d"][b], d["y_d"][b], up[b], yp[b])
    print("  instance %d: rel err u %.2e cost %.2e (oracle %s, rank %d, cost %.6g)" % (
        b, np.max(np.abs(u[b] - ref["optimal_u"])) / np.max(np.abs(ref["optimal_u"])), abs(cost[b] - ref["cost"]) / abs(ref["cost"]),
        ref["status"], ref["rank"], ref["cost"]))
