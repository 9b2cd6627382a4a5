"""Dev tool: per-phase shader-clock breakdown of the cold-solve kernel (diagnostic stamps).

    python tools/phase_stamps.py [batch] [gram_mode] [L] [N]      (defaults 4096 0 30 400; configs[3]: 1024 0 60 1000)"""
import sys
import numpy as np
sys.path.insert(0, ".")
from direct_data_driven_mpc_amd import _lib as L
from direct_data_driven_mpc_amd.engine import BatchedDDMPC
from direct_data_driven_mpc_amd.harness import controller_params, generate_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
gram = int(sys.argv[2]) if len(sys.argv) > 2 else 0
LH = int(sys.argv[3]) if len(sys.argv) > 3 else 30
NN = int(sys.argv[4]) if len(sys.argv) > 4 else 400
cfg = controller_params(dict(L=LH, N=NN)) if LH != 30 else controller_params()
d = generate_batch(range(B), N=NN)
n = 4
up = d["u_d"][:, -n:, :].reshape(B, -1).copy(); yp = d["y_d"][:, -n:, :].reshape(B, -1).copy()
eng = BatchedDDMPC(n=4, m=2, p=2, L_=LH, N=NN, Q=cfg["Q"], R=cfg["R"], u_s=cfg["u_s"], y_s=cfg["y_s"], batch=B,
                   controller_type=L.ROBUST, slack_type=L.SLACK_NONE, eps_max=cfg["eps_max"], lamb_alpha=cfg["lamb_alpha"],
                   lamb_sigma=cfg["lamb_sigma"], c=cfg["c"], gram_mode=gram)
eng.set_data(d["u_d"], d["y_d"])
eng.solve(up, yp)
eng.debug_stamps(True)
eng.solve(up, yp)
st = eng.debug_stamps(False, fetch=True).astype(np.int64)
names = ["staging+tables", "lag blocks (4x4x4)", "base tiles + walks", "fix-up + cholesky", "y + back substitution (+ refinement)", "-", "-", "-"]
dt = np.diff(st[:, :8], axis=1)
tot = st[:, 14] - st[:, 0]
real = (st[:, 13] - st[:, 15]) * 10.0   # ns
print("kernel", eng.kernel_name(), "B", B, "gram_mode", gram)
for i, nm in enumerate(names[:5]):
    col = dt[:, i]
    print("%-38s median %8.0f  p10 %8.0f  p90 %8.0f cycles" % (nm, np.median(col), np.percentile(col, 10), np.percentile(col, 90)))
print("%-38s median %8.0f cycles" % ("outputs + AUTO refinement trigger", np.median(st[:, 14] - st[:, 6])))
print("%-16s median %8.0f cycles; real time median %.1f us; clock %.2f GHz" % ("total", np.median(tot), np.median(real) / 1e3, np.median(tot / np.maximum(real, 1))))
ph = st[:, 7:12]
for i, nm in enumerate(["chol: panel wave in-tile factorisation (+ its diag updates)", "chol: wait at barrier B", "chol: TRSM (panel wave: none)", "chol: wait at barriers A1+A2", "chol: next-diagonal update"]):
    print("%-60s median %8.0f cycles (sum over steps, wave 0)" % (nm, np.median(ph[:, i])))
print("span first entry -> last exit: %.1f us" % ((st[:, 13].max() - st[:, 15].min()) * 10.0 / 1e3))
