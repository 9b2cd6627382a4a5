"""Dev timing: nominal scheme on exact data (every instance goes through the rank-revealing rescue kernel)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from direct_data_driven_mpc_amd import _lib as L
from direct_data_driven_mpc_amd.engine import BatchedDDMPC
from direct_data_driven_mpc_amd.harness import FOUR_TANK, generate_batch
plant = dict(FOUR_TANK); plant["eps_max"] = 0.0
B = 4096
d = generate_batch(range(B), N=400, plant=plant)
A, Bm, Cm, D = (FOUR_TANK[k] for k in "ABCD")
u_s = np.array([1.0, 1.0]); y_s = (Cm @ np.linalg.inv(np.eye(4) - A) @ Bm + D) @ u_s
eng = BatchedDDMPC(n=4, m=2, p=2, L_=30, N=400, Q=3.0, R=1e-4, u_s=u_s, y_s=y_s, batch=B, controller_type=L.NOMINAL)
up = d["u_d"][:, -4:, :].reshape(B, -1).copy(); yp = d["y_d"][:, -4:, :].reshape(B, -1).copy()
eng.set_data(d["u_d"], d["y_d"])
eng.solve(up, yp)
t = time.perf_counter()
for _ in range(5):
    u, c, s, it = eng.solve(up, yp)
dt = (time.perf_counter() - t) / 5
print("nominal, exact data, B=%d: %.2f ms per batch incl. host copies -> %.3e solves/s; status %s" % (B, dt * 1e3, B / dt, sorted(set(s.tolist()))))
