"""Dev tool (GPU): cold-solve rate of the register-resident kernels in both Gram modes for plants with m + p != 4.

    python tools/gram_modes_time.py [--batch 4096] [--steps 10]

DENSE: G = H H' by MFMA inside the kernel (r^2 c / 2 multiply-adds).  STRUCTURED (= AUTO): for two and four channels in the
kernel; otherwise a launch ahead of it forms the tiles -- "structured": the streaming matrix-pipe launch (rr2_gram_tiles*_kernel,
round 5, default), "staged": round 4's ddmpc_gram_tiles_kernel (DDMPC_OPT_GRAM_LAUNCH)."""
import argparse, sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
from direct_data_driven_mpc_amd import _lib as L
from direct_data_driven_mpc_amd.engine import BatchedDDMPC
from direct_data_driven_mpc_amd.harness import generate_batch

ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=4096); ap.add_argument("--steps", type=int, default=10)
a = ap.parse_args()
dev = torch.device("cuda", 0)
for tag, m, p, n, Lh, N in (("SISO        ", 1, 1, 4, 64, 400), ("3 in, 2 out ", 3, 2, 4, 24, 400), ("2 in, 1 out ", 2, 1, 4, 41, 400),
                            ("3 in, 3 out ", 3, 3, 4, 18, 400), ("4 in, 3 out ", 4, 3, 4, 15, 400),
                            ("4 in, 4 out ", 4, 4, 4, 13, 400), ("6 in, 6 out ", 6, 6, 2, 9, 400), ("four-tank   ", 2, 2, 4, 30, 400)):
    rng = np.random.default_rng(5)
    A = rng.normal(size=(n, n)); A *= 0.85 / max(abs(np.linalg.eigvals(A)))
    plant = dict(A=A, B=rng.normal(size=(n, m)), C=rng.normal(size=(p, n)), D=np.zeros((p, m)), eps_max=0.002)
    B = a.batch
    d = generate_batch(range(B), N=N, plant=plant)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    ud, yd = t(d["u_d"]), t(d["y_d"])
    up, yp = t(d["u_d"][:, -n:, :].reshape(B, -1)), t(d["y_d"][:, -n:, :].reshape(B, -1))
    for refine in ("off", "auto"):
        rate = {}; us = {}
        for mode, name in ((L.GRAM_DENSE, "dense"), (L.GRAM_STRUCTURED, "structured"), (L.GRAM_STRUCTURED, "staged")):
            if name == "staged" and m + p in (2, 4):
                continue
            with BatchedDDMPC(n=n, m=m, p=p, L_=Lh, N=N, Q=3.0, R=1e-4, u_s=np.zeros(m), y_s=0.3 * np.ones(p), batch=B, controller_type=L.ROBUST,
                              slack_type=L.SLACK_NONE, eps_max=0.002, lamb_alpha=50.0, lamb_sigma=1000.0, c=1.0, gram_mode=mode) as eng:
                eng.set_refinement(refine)
                if name == "staged":
                    eng.set_gram_launch("staged")
                eng.set_data(ud, yd)
                out = eng.solve(up, yp)
                eng.solve(up, yp, *out)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(a.steps):
                    eng.solve(up, yp, *out)
                torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
                rate[name] = B / dt; us[name] = out[0].cpu().numpy().copy()
                kern = eng.kernel_name()
        diff = np.max(np.abs(us["dense"] - us["structured"])) / np.max(np.abs(us["dense"]))
        extra = "" if "staged" not in rate else "; staged launch %.3e (x %.2f), diff %.1e" % (
            rate["staged"], rate["staged"] / rate["dense"], np.max(np.abs(us["staged"] - us["structured"])) / np.max(np.abs(us["dense"])))
        print("%s m=%d p=%d r=%3d c=%3d %s B=%d refinement %-4s: dense %.3e solves/s, structured %.3e solves/s (x %.2f), max rel diff in u %.1e%s" % (
            tag, m, p, (m + p) * (Lh + n), N - Lh - n + 1, kern, B, refine, rate["dense"], rate["structured"], rate["structured"] / rate["dense"], diff, extra), flush=True)
