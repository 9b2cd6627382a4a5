"""Dev tool (GPU): wall time of ddmpc_prepare on the benchmark batch under the three refinement modes."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
from direct_data_driven_mpc_amd import _lib as L
from direct_data_driven_mpc_amd.engine import BatchedDDMPC
from direct_data_driven_mpc_amd.harness import controller_params, generate_batch
cfg = controller_params()
B = 4096
d = generate_batch(range(B), N=cfg["N"])
n, m, p = cfg["n"], cfg["m"], cfg["p"]
dev = torch.device("cuda", 0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
eng = BatchedDDMPC(n=n, m=m, p=p, L_=cfg["L"], N=cfg["N"], Q=cfg["Q"], R=cfg["R"], u_s=cfg["u_s"], y_s=cfg["y_s"], batch=B,
                   controller_type=L.ROBUST, slack_type=L.SLACK_NONE, eps_max=cfg["eps_max"], lamb_alpha=cfg["lamb_alpha"],
                   lamb_sigma=cfg["lamb_sigma"], c=cfg["c"])
eng.set_data(t(d["u_d"]), t(d["y_d"]))
for mode in ("off", "auto", "always", "auto", "off"):
    ts = []
    for _ in range(4):
        eng.set_refinement(mode)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.prepare()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(mode, ["%.2f" % x for x in ts], flush=True)
