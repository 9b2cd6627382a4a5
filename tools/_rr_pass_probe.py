import sys
import numpy as np
sys.path.insert(0, ".")
from direct_data_driven_mpc_amd import _lib as L
from direct_data_driven_mpc_amd.engine import BatchedDDMPC
from direct_data_driven_mpc_amd.harness import generate_batch
rng = np.random.default_rng(0)
ns = n = 8; m = p = 8; Lh = 30; N = 2000
A = rng.normal(size=(ns, ns)); A *= 0.9 / max(abs(np.linalg.eigvals(A)))
plant = dict(A=A, B=rng.normal(size=(ns, m)), C=rng.normal(size=(p, ns)), D=np.zeros((p, m)), eps_max=0.0)
u_s = 0.1 * np.ones(m); y_s = (plant["C"] @ np.linalg.inv(np.eye(ns) - A) @ plant["B"]) @ u_s
for B in (64, 512):
    d = generate_batch(range(B), N=N, plant=plant)
    up = d["u_d"][:, -n:, :].reshape(B, -1).copy(); yp = d["y_d"][:, -n:, :].reshape(B, -1).copy()
    eng = BatchedDDMPC(n=n, m=m, p=p, L_=Lh, N=N, Q=3.0, R=1e-4, u_s=u_s, y_s=y_s, batch=B, controller_type=L.NOMINAL)
    eng.set_data(d["u_d"], d["y_d"])
    eng.prepare(); eng.step(up, yp)
    eng.debug_stamps(True)
    eng.step(up, yp)
    st = eng.debug_stamps(False, fetch=True).astype(np.int64).reshape(-1, 16)[:B]
    seq = [6, 8, 9, 10, 11, 12, 13, 14, 15, 7]
    names = ["back(G,nlive)", "H(H'x) + perm", "rows(C)+cols(L_RF)+back(G,nF)", "H(H'x) #2", "fwd(G,nlive)", "fwd(G,nF)",
             "rows(L_RF)+cols(C)+fwd(T)", "back(T)", "rest (later passes + outputs)"]
    print("B", B)
    for i, nm in enumerate(names):
        print("  %-34s %8.1f us" % (nm, np.median((st[:, seq[i + 1]] - st[:, seq[i]]) / 100.0)))
    eng.close()
