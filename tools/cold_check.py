"""Dev tool: parity + timing + phase stamps of the cold-solve kernel on the headline configuration."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
from direct_data_driven_mpc_amd import _lib as L
from direct_data_driven_mpc_amd.engine import BatchedDDMPC
from direct_data_driven_mpc_amd.harness import controller_params, generate_batch
from oracle import ddmpc_oracle as orc

def engine(spec, N, B, **kw):
    return BatchedDDMPC(n=spec.n, m=spec.m, p=spec.p, L_=spec.L, N=N, Q=spec.Q, R=spec.R, u_s=spec.u_s, y_s=spec.y_s, batch=B,
                        controller_type=L.ROBUST if spec.robust else L.NOMINAL,
                        slack_type=L.SLACK_CONVEX if spec.slack == "convex" else L.SLACK_NONE, eps_max=spec.eps_max,
                        lamb_alpha=spec.lamb_alpha, lamb_sigma=spec.lamb_sigma, c=spec.c, use_terminal_constraint=spec.tec, **kw)

def parity(tag, B, N=400, **kw):
    spec = orc.spec_from_params(N=N, **kw)
    d = generate_batch(range(B), N=N)
    n = spec.n
    up = d["u_d"][:, -n:, :].reshape(B, -1).copy(); yp = d["y_d"][:, -n:, :].reshape(B, -1).copy()
    with engine(spec, N, B) as eng:
        eng.set_data(d["u_d"], d["y_d"])
        u, cost, st, it = eng.solve(up, yp)
        wu = wc = 0.0
        for b in range(B):
            sol = orc.solve_fullspace(spec, d["u_d"][b], d["y_d"][b], up[b], yp[b])
            wu = max(wu, np.max(np.abs(u[b] - sol.optimal_u)) / np.max(np.abs(sol.optimal_u)))
            wc = max(wc, abs(cost[b] - sol.cost) / max(abs(sol.cost), 1e-9))
        print("%-14s %s B=%d status %s iters %s  err u %.2e cost %.2e" % (tag, eng.kernel_name(), B, sorted(set(st.tolist())), sorted(set(it.tolist())), wu, wc), flush=True)
        if spec.robust:
            sol = orc.solve_fullspace(spec, d["u_d"][0], d["y_d"][0], up[0], yp[0])
            a = eng.get_solution("alpha"); sg = eng.get_solution("sigma")
            print("    alpha err %.2e sigma err %.2e" % (np.max(np.abs(a[0] - sol.alpha)), np.max(np.abs(sg[0] - sol.sigma))))
            uw, cw, sw, _ = eng.step(up, yp)
            print("    warm step vs cold: %.2e" % (np.max(np.abs(uw - u)) / np.max(np.abs(u))))

def timing(B=4096, slack=0, stamps=True, refine="auto"):
    cfg = controller_params(dict(slack_var_constraint_type=slack))
    spec = orc.spec_from_params(slack_var_constraint_type=slack)
    d = generate_batch(range(B))
    dev = torch.device("cuda", 0)
    ud, yd = torch.from_numpy(d["u_d"]).to(dev), torch.from_numpy(d["y_d"]).to(dev)
    up = torch.from_numpy(d["u_d"][:, -4:, :].reshape(B, -1).copy()).to(dev)
    yp = torch.from_numpy(d["y_d"][:, -4:, :].reshape(B, -1).copy()).to(dev)
    with engine(spec, 400, B) as eng:
        eng.set_refinement(refine)
        eng.set_data(ud, yd)
        out = eng.solve(up, yp)
        for _ in range(5): eng.solve(up, yp, *out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): eng.solve(up, yp, *out)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 100
        print("timing slack=%d B=%d: %.1f us per launch, %.3e solves/s, status ok %s" % (slack, B, ms * 1e3, B / ms * 1e3, bool((out[2] == 0).all())), flush=True)
        if not stamps:
            return
        eng.debug_stamps(True)
        eng.solve(up, yp, *out)
        st = eng.debug_stamps(False, fetch=True).astype(np.int64)
        dt = np.diff(st[:, :7], axis=1)
        names = ["staging+tables", "lag blocks", "base+walks", "fixup+cholesky", "y + back subst", "-"]
        for i, nm in enumerate(names[:5]):
            print("   %-16s median %8.0f cycles" % (nm, np.median(dt[:, i])))
        print("   total %8.0f cycles" % np.median(st[:, 14] - st[:, 0]))
        for i, nm in enumerate(["F factor (wave 0, incl. U)", "wait B", "T trsm", "wait A1+A2", "U diag updates (in F)"]):
            print("   chol %-30s median %8.0f cycles (sum over steps, wave 0)" % (nm, np.median(st[:, 7 + i])))

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "ref":         # refinement modes, in-process
        for rep in range(3):
            for ref in ("off", "auto", "always"):
                print("refine=%-6s" % ref, end="  ")
                timing(4096, 0, stamps=False, refine=ref)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "occ":         # phase stamps at 1, 2 and 3 workgroups per CU
        for B in (256, 512, 768, 4096):
            timing(B, 0)
        sys.exit(0)
    parity("robust/none", 8)
    parity("robust/convex", 8, slack_var_constraint_type=1)
    parity("ucon", 4, tec=False)
    parity("nominal", 4, controller_type=0)
    parity("N=401 ragged", 4, N=401)
    timing(4096, 0)
    timing(4096, 1)
