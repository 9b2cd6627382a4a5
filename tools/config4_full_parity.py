"""BASELINE configs[3] (robust, L = 60, N = 1000, B = 1024) at full batch: every instance checked against the compiled CPU
restatement (oracle/ddmpc_oracle_c.c), slack NONE and CONVEX.

    python tools/config4_full_parity.py
"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import bench
from oracle import oracle_c
from direct_data_driven_mpc_amd.harness import controller_params, generate_batch

B, Lh, N = 1024, 60, 1000
d = generate_batch(range(B), N=N)
refs = {}
for slack in (0, 1):
    cfg = controller_params(dict(L=Lh, N=N, slack_var_constraint_type=slack))
    n = cfg["n"]
    up = d["u_d"][:, -n:, :].reshape(B, -1).copy(); yp = d["y_d"][:, -n:, :].reshape(B, -1).copy()
    t0 = time.perf_counter()
    u_ref, c_ref, st_ref, it_ref = oracle_c.solve_batch(bench._oracle_spec(cfg), N, d["u_d"], d["y_d"], up, yp, threads=bench.host_cores())
    assert not np.count_nonzero(st_ref)
    refs[slack] = (cfg, up, yp, u_ref, c_ref)
    print("slack %d: C restatement %.0f solves/s on %d threads (%.1f s)" % (slack, B / (time.perf_counter() - t0), bench.host_cores(), time.perf_counter() - t0), flush=True)

import torch
from direct_data_driven_mpc_amd import _lib as L
from direct_data_driven_mpc_amd.engine import BatchedDDMPC
dev = torch.device("cuda", 0)
for slack in (0, 1):
    cfg, up, yp, u_ref, c_ref = refs[slack]
    eng = BatchedDDMPC(n=4, m=2, p=2, L_=Lh, N=N, Q=cfg["Q"], R=cfg["R"], u_s=cfg["u_s"], y_s=cfg["y_s"], batch=B,
                       controller_type=L.ROBUST, slack_type=L.SLACK_CONVEX if slack else L.SLACK_NONE, eps_max=cfg["eps_max"],
                       lamb_alpha=cfg["lamb_alpha"], lamb_sigma=cfg["lamb_sigma"], c=cfg["c"])
    t = lambda x: torch.from_numpy(x).to(dev)
    eng.set_data(t(d["u_d"]), t(d["y_d"]))
    out = eng.solve(t(up), t(yp)); torch.cuda.synchronize()
    u = out[0].cpu().numpy(); c = out[1].cpu().numpy(); st = out[2].cpu().numpy(); it = out[3].cpu().numpy()
    eu = np.max(np.max(np.abs(u - u_ref), axis=1) / np.max(np.abs(u_ref), axis=1)); ec = np.max(np.abs(c - c_ref) / np.abs(c_ref))
    print("cfg4 slack %s, kernel %s: %d instances, max rel err u %.3e (tol 1e-8) cost %.3e (tol 1e-9), non-optimal %d, iters mean %.2f" % (
        "CONVEX" if slack else "NONE", eng.kernel_name(), B, eu, ec, int(np.count_nonzero(st)), it.mean()), flush=True)
    eng.close()
