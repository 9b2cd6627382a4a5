"""Dev timing: batch-size sweep of the two global-workspace kernels at the cfg-5 size (608 rows): how the time per
batch grows from 64 to 1024 instances shows how many workgroups share a CU (256 CUs: flat up to 256, x1.5 at 512 =
two per CU, x2 from 512 to 1024 = two rounds)."""
import sys, time, numpy as np
sys.path.insert(0, ".")
from direct_data_driven_mpc_amd.harness import generate_batch
from direct_data_driven_mpc_amd import _lib as L
from direct_data_driven_mpc_amd.engine import BatchedDDMPC
rng = np.random.default_rng(0)
ns = n = 8; m = p = 8; Lh = 30; N = 2000
A = rng.normal(size=(ns, ns)); A *= 0.9 / max(abs(np.linalg.eigvals(A)))
plant = dict(A=A, B=rng.normal(size=(ns, m)), C=rng.normal(size=(p, ns)), D=np.zeros((p, m)), eps_max=0.0)
u_s = 0.1 * np.ones(m); y_s = (plant["C"] @ np.linalg.inv(np.eye(ns) - A) @ plant["B"]) @ u_s
dall = generate_batch(range(1024), N=N, plant=plant)
for robust in (False, True):
    for B in (64, 128, 256, 512, 1024):
        d = {k: dall[k][:B] for k in ("u_d", "y_d")}
        up = d["u_d"][:, -n:, :].reshape(B, -1).copy(); yp = d["y_d"][:, -n:, :].reshape(B, -1).copy()
        kw = dict(controller_type=L.ROBUST, slack_type=L.SLACK_CONVEX, eps_max=0.002, lamb_alpha=50.0, lamb_sigma=1000.0, c=1.0) if robust else dict(controller_type=L.NOMINAL)
        eng = BatchedDDMPC(n=n, m=m, p=p, L_=Lh, N=N, Q=3.0, R=1e-4, u_s=u_s, y_s=y_s, batch=B, **kw)
        eng.set_data(d["u_d"], d["y_d"])
        eng.solve(up, yp)
        t = time.perf_counter(); eng.solve(up, yp); dt = time.perf_counter() - t
        print("%s B=%4d: %.2f ms" % (eng.kernel_name(), B, dt * 1e3), flush=True)
        eng.close()
