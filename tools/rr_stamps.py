"""Dev tool: per-phase breakdown of ddmpc_nominal_rr_kernel at the cfg-5 size (in-kernel s_memrealtime stamps, 100 MHz,
median over the instances of the batch) at several batch sizes: 128 / 256 = at most one workgroup per CU, 512 = two.

    python tools/rr_stamps.py [batch ...]
"""
import sys
import numpy as np
sys.path.insert(0, ".")
import torch
from direct_data_driven_mpc_amd import _lib as L
from direct_data_driven_mpc_amd.engine import BatchedDDMPC
from direct_data_driven_mpc_amd.harness import generate_batch

batches = [int(x) for x in sys.argv[1:]] or [256, 512]
rng = np.random.default_rng(0)
ns = n = 8; m = p = 8; Lh = 30; N = 2000
A = rng.normal(size=(ns, ns)); A *= 0.9 / max(abs(np.linalg.eigvals(A)))
plant = dict(A=A, B=rng.normal(size=(ns, m)), C=rng.normal(size=(p, ns)), D=np.zeros((p, m)), eps_max=0.0)
u_s = 0.1 * np.ones(m); y_s = (plant["C"] @ np.linalg.inv(np.eye(ns) - A) @ plant["B"]) @ u_s
names = ["gram", "cholG", "fwd+z0", "T form", "cholT", "solve", "out"]
for B in batches:
    d = generate_batch(range(B), N=N, plant=plant)
    up = d["u_d"][:, -n:, :].reshape(B, -1).copy(); yp = d["y_d"][:, -n:, :].reshape(B, -1).copy()
    eng = BatchedDDMPC(n=n, m=m, p=p, L_=Lh, N=N, Q=3.0, R=1e-4, u_s=u_s, y_s=y_s, batch=B, controller_type=L.NOMINAL)
    eng.set_data(d["u_d"], d["y_d"])
    eng.solve(up, yp)
    eng.debug_stamps(True)
    eng.solve(up, yp)
    st = eng.debug_stamps(False, fetch=True).astype(np.int64).reshape(-1, 8)[:B]      # 8 stamps per instance
    print("B = %d" % B)
    for i, nm in enumerate(names):
        print("   %-8s median %9.1f us" % (nm, np.median((st[:, i + 1] - st[:, i]) / 100.0)))
    print("   total    median %9.1f us; start spread %.1f us, last end %.1f us after the first start" % (
        np.median((st[:, 7] - st[:, 0]) / 100.0), (st[:, 0].max() - st[:, 0].min()) / 100.0, (st[:, 7].max() - st[:, 0].min()) / 100.0))
    eng.close()
