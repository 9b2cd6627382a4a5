"""Dev tool: PCIe-inclusive time of one batch from host pointers -- ddmpc_solve_from_host (chunked upload on a copy stream
overlapped with the solves) against ddmpc_set_data + ddmpc_solve, per refinement mode, in one process."""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np
import torch; torch.cuda.init()
from direct_data_driven_mpc_amd import _lib as L
from direct_data_driven_mpc_amd.engine import BatchedDDMPC
from direct_data_driven_mpc_amd.harness import controller_params, generate_batch
B = 4096
cfg = controller_params()
d = generate_batch(range(B))
up = d["u_d"][:, -4:, :].reshape(B, -1).copy(); yp = d["y_d"][:, -4:, :].reshape(B, -1).copy()
for ref in ("off", "auto", "off", "auto"):
    eng = BatchedDDMPC(n=4, m=2, p=2, L_=30, N=400, Q=cfg["Q"], R=cfg["R"], u_s=cfg["u_s"], y_s=cfg["y_s"], batch=B,
                       controller_type=L.ROBUST, slack_type=L.SLACK_NONE, eps_max=cfg["eps_max"], lamb_alpha=cfg["lamb_alpha"],
                       lamb_sigma=cfg["lamb_sigma"], c=cfg["c"])
    eng.set_refinement(ref)
    eng.solve_from_host(d["u_d"], d["y_d"], up, yp)
    ts = []
    for _ in range(8):
        t0 = time.perf_counter(); eng.solve_from_host(d["u_d"], d["y_d"], up, yp); ts.append(time.perf_counter() - t0)
    t0 = time.perf_counter(); eng.set_data(d["u_d"], d["y_d"]); eng.solve(up, yp); t1 = time.perf_counter() - t0
    print("refine=%s pipelined: %s ms; set_data+solve %.2f ms" % (ref, " ".join("%.2f" % (t * 1e3) for t in ts), t1 * 1e3), flush=True)
    eng.close()
