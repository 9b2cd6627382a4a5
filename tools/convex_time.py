"""Dev tool (GPU): slack CONVEX on BASELINE configs[1] / configs[3] -- rank-k update of the kept factor (round 5) against a new
factorisation per active-set iteration (rounds 1-4), same process, same data; slack NONE beside them for scale."""
import sys
import numpy as np
sys.path.insert(0, ".")
import torch
from direct_data_driven_mpc_amd import _lib as L
from direct_data_driven_mpc_amd.engine import BatchedDDMPC
from direct_data_driven_mpc_amd.harness import controller_params, generate_batch


def run(tag, B, Lh, N, slack, upd, reps=50):
    cfg = controller_params(dict(L=Lh, N=N, slack_var_constraint_type=slack))
    d = generate_batch(range(B), N=N)
    dev = torch.device("cuda", 0)
    n = cfg["n"]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    ud, yd = t(d["u_d"]), t(d["y_d"])
    up, yp = t(d["u_d"][:, -n:, :].reshape(B, -1)), t(d["y_d"][:, -n:, :].reshape(B, -1))
    with BatchedDDMPC(n=n, m=cfg["m"], p=cfg["p"], L_=Lh, N=N, Q=cfg["Q"], R=cfg["R"], u_s=cfg["u_s"], y_s=cfg["y_s"], batch=B,
                      controller_type=L.ROBUST, slack_type=L.SLACK_CONVEX if slack else L.SLACK_NONE, eps_max=cfg["eps_max"],
                      lamb_alpha=cfg["lamb_alpha"], lamb_sigma=cfg["lamb_sigma"], c=cfg["c"]) as eng:
        eng.set_convex_update(upd)
        eng.set_data(ud, yd)
        out = eng.solve(up, yp)
        for _ in range(5):
            eng.solve(up, yp, *out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            eng.solve(up, yp, *out)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        it = out[3].cpu().numpy()
        if slack and "--stamps" in sys.argv:
            eng.debug_stamps(True)
            eng.solve(up, yp, *out)
            st = eng.debug_stamps(False, fetch=True).astype(np.int64)
            two = it == 2
            tot = st[:, 14] - st[:, 0]
            print("   cycles per workgroup: 1 iteration %.0f | 2 iterations %.0f; of the second (update=%s): forward %.0f  k x k %.0f  back %.0f"
                  % (np.median(tot[~two]), np.median(tot[two]), upd, np.median(st[two, 9]), np.median(st[two, 10]), np.median(st[two, 11])), flush=True)
        print("%-22s %s B=%5d  %7.1f us  %.3e solves/s  iters mean %.3f max %d  non-optimal %d" % (
            tag, eng.kernel_name(), B, ms * 1e3, B / ms * 1e3, it.mean(), it.max(), int((out[2] != 0).sum().item())), flush=True)
        return out[0].cpu().numpy(), it


if __name__ == "__main__":
    for (B, Lh, N, name) in ((4096, 30, 400, "cfg2"), (1024, 60, 1000, "cfg4")):
        if "--only" in sys.argv and sys.argv[sys.argv.index("--only") + 1] != name:
            continue
        run(name + " NONE", B, Lh, N, 0, True)
        ua, ia = run(name + " CONVEX update", B, Lh, N, 1, True)
        ub, ib = run(name + " CONVEX refactor", B, Lh, N, 1, False)
        print("   update vs refactor: iters equal %s, max rel diff u %.2e" % (np.array_equal(ia, ib), np.max(np.abs(ua - ub)) / np.max(np.abs(ub))), flush=True)
