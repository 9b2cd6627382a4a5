"""Dev tool: accuracy and time of the cfg-5 solve (ddmpc_nominal_rr_kernel) against the cap on its refinement passes, per
instance: which instances take a second pass, and what it buys them (yardstick: the model-based solution).

    python tools/cfg5_passes.py [ninst_checked]
"""
import sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
import test_gpu_round3 as T
from oracle.nominal_exact import solve_nominal_model_based

K = int(sys.argv[1]) if len(sys.argv) > 1 else 128
B = 512
spec, plant, N, d, up, yp = T._config5(B)
refs = [solve_nominal_model_based(spec, plant, up[b], yp[b]) for b in range(K)]
dev = torch.device("cuda", 0)
t = lambda x: torch.from_numpy(x).to(dev)
res = {}
with T._spec_engine(spec, N, B) as eng:
    eng.set_data(t(d["u_d"]), t(d["y_d"]))
    upd, ypd = t(up), t(yp)
    for mp in (1, 2, 3):
        eng.set_refinement("auto", max_passes=mp)
        out = eng.solve(upd, ypd); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            eng.solve(upd, ypd, *out)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        u = out[0].cpu().numpy(); c = out[1].cpu().numpy()
        eu = np.array([np.max(np.abs(u[b] - refs[b]["optimal_u"])) / np.max(np.abs(refs[b]["optimal_u"])) for b in range(K)])
        ec = np.array([abs(c[b] - refs[b]["cost"]) / abs(refs[b]["cost"]) for b in range(K)])
        res[mp] = (u.copy(), eu, ec)
        print("max passes %d: %.2f ms per batch; first %d instances vs the model-based solution: u max %.2e median %.2e, cost max %.2e" % (
            mp, ms, K, eu.max(), np.median(eu), ec.max()), flush=True)
u1, e1, _ = res[1]; u3, e3, _ = res[3]
chg = np.max(np.abs(u1 - u3), axis=1) / np.max(np.abs(u3), axis=1)
took = chg > 0
print("instances whose result changes with more than one pass allowed: %d of %d" % (took.sum(), B))
kk = np.nonzero(took[:K])[0]
if kk.size:
    print("  among the checked ones (%d): error with 1 pass  max %.2e median %.2e;  with up to 3 passes  max %.2e median %.2e;  change max %.2e" % (
        kk.size, e1[kk].max(), np.median(e1[kk]), e3[kk].max(), np.median(e3[kk]), chg[kk].max()))
