"""Dev tool (GPU): cold solves of data sets whose trajectory is longer than the cold-solve kernel's LDS holds (four-tank,
L = 30, N = 6000 / 20000): streaming Gram + tile packing + the kernel in `gpre` mode + the streamed residual check, per batch."""
import sys, time
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
from direct_data_driven_mpc_amd import _lib as L
from direct_data_driven_mpc_amd.engine import BatchedDDMPC
from direct_data_driven_mpc_amd.harness import controller_params, generate_batch

for N, B in ((400, 1024), (3000, 1024), (6000, 1024), (20000, 256)):
    cfg = controller_params(dict(N=N))
    d = generate_batch(range(B), N=N)
    dev = torch.device("cuda", 0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    ud, yd = t(d["u_d"]), t(d["y_d"])
    up, yp = t(d["u_d"][:, -4:, :].reshape(B, -1)), t(d["y_d"][:, -4:, :].reshape(B, -1))
    with BatchedDDMPC(n=4, m=2, p=2, L_=30, N=N, Q=cfg["Q"], R=cfg["R"], u_s=cfg["u_s"], y_s=cfg["y_s"], batch=B, controller_type=L.ROBUST,
                      slack_type=L.SLACK_NONE, eps_max=cfg["eps_max"], lamb_alpha=cfg["lamb_alpha"], lamb_sigma=cfg["lamb_sigma"], c=cfg["c"]) as eng:
        eng.set_data(ud, yd)
        out = eng.solve(up, yp)
        for _ in range(3):
            eng.solve(up, yp, *out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            eng.solve(up, yp, *out)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        gb = B * N * 4 * 8 / 1e9
        print("N=%6d B=%5d: %8.3f ms per batch, %.3e solves/s, trajectory %.2f GB -> %.0f GB/s of data read; statuses %s" % (
            N, B, ms, B / ms * 1e3, gb, gb / ms * 1e3, sorted(set(out[2].cpu().numpy().tolist()))), flush=True)
