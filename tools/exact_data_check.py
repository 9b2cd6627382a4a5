"""Dev check: nominal controller on NOISE-FREE data (rank-deficient Gram): what does the engine report?"""
import sys
import numpy as np
sys.path.insert(0, ".")
from direct_data_driven_mpc_amd import _lib as L
from direct_data_driven_mpc_amd.engine import BatchedDDMPC
from direct_data_driven_mpc_amd.harness import FOUR_TANK, generate_batch
from oracle import ddmpc_oracle as orc

plant = dict(FOUR_TANK); plant["eps_max"] = 0.0
B = 8
d = generate_batch(range(B), N=400, plant=plant)
for robust in (False, True):
    spec = orc.spec_from_params(controller_type=1 if robust else 0)
    up = d["u_d"][:, -4:, :].reshape(B, -1).copy(); yp = d["y_d"][:, -4:, :].reshape(B, -1).copy()
    eng = BatchedDDMPC(n=4, m=2, p=2, L_=30, N=400, Q=spec.Q, R=spec.R, u_s=spec.u_s, y_s=spec.y_s, batch=B,
                       controller_type=L.ROBUST if robust else L.NOMINAL, slack_type=L.SLACK_NONE,
                       eps_max=spec.eps_max, lamb_alpha=spec.lamb_alpha, lamb_sigma=spec.lamb_sigma, c=spec.c)
    eng.set_data(d["u_d"], d["y_d"])
    u, cost, status, it = eng.solve(up, yp)
    print("robust" if robust else "nominal", "status", status.tolist(), "cost", cost[:3])
    for b in range(2):
        try:
            sol = orc.solve_fullspace(spec, d["u_d"][b], d["y_d"][b], up[b], yp[b])
        except np.linalg.LinAlgError as e:
            print("   oracle failed:", e); continue
        print("   oracle", sol.status, sol.cost, "u0", sol.optimal_u[:2], "gpu u0", u[b, :2],
              "rel err", np.max(np.abs(u[b] - sol.optimal_u)) / np.max(np.abs(sol.optimal_u)))
    H = np.vstack([orc.hankel_matrix(d["u_d"][0], 34), orc.hankel_matrix(d["y_d"][0], 34)])
    sv = np.linalg.svd(H, compute_uv=False)
    print("   rank(H) numeric", int(np.sum(sv > sv[0] * 1e-10)), "of", H.shape[0], "sv tail", sv[70:75])
    eng.close()
