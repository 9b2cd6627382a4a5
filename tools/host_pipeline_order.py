import os, sys, time
sys.path.insert(0, ".")
import numpy as np
import torch
from direct_data_driven_mpc_amd import _lib as L
from direct_data_driven_mpc_amd.engine import BatchedDDMPC
from direct_data_driven_mpc_amd.harness import controller_params, generate_batch
B = 4096
cfg = controller_params()
def mk():
    return BatchedDDMPC(n=4, m=2, p=2, L_=30, N=400, Q=cfg["Q"], R=cfg["R"], u_s=cfg["u_s"], y_s=cfg["y_s"], batch=B,
                       controller_type=L.ROBUST, slack_type=L.SLACK_NONE, eps_max=cfg["eps_max"], lamb_alpha=cfg["lamb_alpha"],
                       lamb_sigma=cfg["lamb_sigma"], c=cfg["c"])
def run(pipelined, fresh, pre_torch=False):
    d = generate_batch(range(B)) if fresh else D
    up = d["u_d"][:, -4:, :].reshape(B, -1).copy(); yp = d["y_d"][:, -4:, :].reshape(B, -1).copy()
    eng = mk()
    if pre_torch:
        dev = torch.device("cuda", 0)
        ud = torch.from_numpy(d["u_d"]).to(dev); yd = torch.from_numpy(d["y_d"]).to(dev)
        eng.set_data(ud, yd); eng.solve(torch.from_numpy(up).to(dev), torch.from_numpy(yp).to(dev)); torch.cuda.synchronize()
    eng.solve_from_host(d["u_d"], d["y_d"], up, yp)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        if pipelined: eng.solve_from_host(d["u_d"], d["y_d"], up, yp)
        else: eng.set_data(d["u_d"], d["y_d"]); eng.solve(up, yp)
        ts.append((time.perf_counter() - t0) * 1e3)
    eng.close()
    print("pipelined=%d fresh_arrays=%d pre_torch=%d: %s ms" % (pipelined, fresh, pre_torch, " ".join("%.2f" % t for t in ts)), flush=True)
D = generate_batch(range(B))
run(True, False); run(False, False); run(True, False); run(True, True); run(False, True); run(True, True)
big = torch.empty(1 << 28, dtype=torch.float64, device="cuda"); del big
run(True, True); run(False, True); run(True, True, pre_torch=True)
