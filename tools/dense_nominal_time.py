"""Dev tool (GPU): NOMINAL controllers at configs[4]'s size with dense weighting matrices on the phase kernels -- time per batch of
512 against the diagonal case, parity of one instance against the model-based solution (run under rocprofv3 for the per-kernel split).

    python tools/dense_nominal_time.py [--steps 5]
"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, "tests")
import numpy as np, torch
from direct_data_driven_mpc_amd import _lib as L
from oracle import ddmpc_oracle as orc
from oracle.nominal_exact import solve_nominal_model_based
from test_gpu_round5 import _spec_engine
from test_gpu_round3 import _config5
ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=5); a = ap.parse_args()
spec5, plant5, N5, d5, up5, yp5 = _config5(512)
rng = np.random.default_rng(5)
def spd(k, s):
    X = rng.normal(size=(k, k)); return s * (np.eye(k) + 0.3 * (X @ X.T) / k)
Q5, R5 = spd(spec5.p * spec5.L, 3.0), spd(spec5.m * spec5.L, 1e-4)
sp5 = orc.QPSpec(n=spec5.n, m=spec5.m, p=spec5.p, L=spec5.L, Q=Q5, R=R5, u_s=spec5.u_s, y_s=spec5.y_s, robust=False, eps_max=0.0,
                 lamb_alpha=0.0, lamb_sigma=0.0, c=0.0, slack="none", tec=True)
dev = torch.device("cuda", 0)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
for tag, sp in (("diagonal", spec5), ("dense", sp5)):
    with _spec_engine(sp, N5, 512) as eng:
        ud, yd, tup, typ = t(d5["u_d"]), t(d5["y_d"]), t(up5), t(yp5)
        eng.set_data(ud, yd)
        o = eng.solve(tup, typ)
        times = []
        for _ in range(a.steps):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            o = eng.solve(tup, typ, *o)
            torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
        dt = sorted(times)[len(times) // 2]                     # (median: the first timed solve of a handle still allocates)
        st = o[2].cpu().numpy(); u0 = o[0][0].cpu().numpy(); c0 = float(o[1][0])
    mod = solve_nominal_model_based(sp, plant5, up5[0], yp5[0])
    print("configs[4] size, %s weights: %.2f ms per 512 cold solves (%.3e/s; per solve %s ms), statuses %s; instance 0 against the model-based solution: u %.2e cost %.2e" % (
        tag, dt * 1e3, 512 / dt, " ".join("%.2f" % (x * 1e3) for x in times), sorted(set(st.tolist())), np.max(np.abs(u0 - mod["optimal_u"])) / np.max(np.abs(mod["optimal_u"])),
        abs(c0 - mod["cost"]) / abs(mod["cost"])), flush=True)
