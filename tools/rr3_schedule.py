"""Dev tool (GPU): when every workgroup of rr3_solve_kernel ran -- start tick and duration per instance from the record the
kernel leaves (ddmpc_debug_workspace) -- for the ROBUST scheme at configs[4]'s size with the slack box."""
import ctypes as C
import sys
import numpy as np
sys.path.insert(0, ".")
import torch
from direct_data_driven_mpc_amd import _lib as L
from direct_data_driven_mpc_amd.engine import BatchedDDMPC
from direct_data_driven_mpc_amd.harness import generate_batch

rng = np.random.default_rng(0)
ns = n = 8; m = p = 8; Lh = 30; N = 2000; B = 512
A = rng.normal(size=(ns, ns)); A *= 0.9 / max(abs(np.linalg.eigvals(A)))
plant = dict(A=A, B=rng.normal(size=(ns, m)), C=rng.normal(size=(p, ns)), D=np.zeros((p, m)), eps_max=0.002)
u_s = 0.1 * np.ones(m); y_s = (plant["C"] @ np.linalg.inv(np.eye(ns) - A) @ plant["B"]) @ u_s
d = generate_batch(range(B), N=N, plant=plant)
dev = torch.device("cuda", 0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
ud, yd = t(d["u_d"]), t(d["y_d"])
up, yp = t(d["u_d"][:, -n:, :].reshape(B, -1)), t(d["y_d"][:, -n:, :].reshape(B, -1))
eng = BatchedDDMPC(n=n, m=m, p=p, L_=Lh, N=N, Q=3.0, R=1e-4, u_s=u_s, y_s=y_s, batch=B, controller_type=L.ROBUST,
                   slack_type=L.SLACK_CONVEX, eps_max=0.002, lamb_alpha=50.0, lamb_sigma=1000.0, c=1.0)
eng.set_data(ud, yd)
out = eng.solve(up, yp)
eng.prepare()
out = eng.step(up, yp, *out)
torch.cuda.synchronize()
lib = L.load()
nm = C.c_int64(0); na = C.c_int64(0)
L.check(lib.ddmpc_debug_workspace(eng._h, 0, None, 0, None, 0, C.byref(na), C.byref(nm)))
rec = np.zeros((B, nm.value), dtype=np.int32)
for b in range(B):
    L.check(lib.ddmpc_debug_workspace(eng._h, b, None, 0, C.c_void_p(rec[b].ctypes.data), nm.value, None, None))
k, st, it = rec[:, 3], rec[:, 1], rec[:, 2]
t0 = rec[:, -4].astype(np.int64); dur = rec[:, -3].astype(np.int64)
t0 = (t0 - t0.min()) & 0x7fffffff
print("k: median %d p90 %d max %d; iterations %s; states %s" % (np.median(k), np.percentile(k, 90), k.max(), np.bincount(it), np.bincount(st)))
print("duration per workgroup (us): min %.0f median %.0f p90 %.0f max %.0f" % tuple(x / 100 for x in (dur.min(), np.median(dur), np.percentile(dur, 90), dur.max())))
print("start (us): first quarter ends %.0f, half %.0f, last start %.0f; last end %.0f" % (np.percentile(t0, 25) / 100, np.percentile(t0, 50) / 100, t0.max() / 100, (t0 + dur).max() / 100))
order = np.argsort(t0)
print("starts (us) of every 32nd workgroup:", " ".join("%.0f" % (t0[order[i]] / 100) for i in range(0, B, 32)))
