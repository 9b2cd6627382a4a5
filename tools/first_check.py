"""Quick GPU sanity check of the cold-solve kernel against the CPU oracle (dev tool)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from direct_data_driven_mpc_amd import _lib as L
from direct_data_driven_mpc_amd.engine import BatchedDDMPC
from oracle import ddmpc_oracle as orc

def run(tag, B, **kw):
    spec = orc.spec_from_params(**kw)
    insts = [orc.generate_instance(s, N=kw.get("N", 400)) for s in range(B)]
    u_d = np.stack([i["u_d"] for i in insts]); y_d = np.stack([i["y_d"] for i in insts])
    n, m, p = spec.n, spec.m, spec.p
    up = u_d[:, -n:, :].reshape(B, -1).copy(); yp = y_d[:, -n:, :].reshape(B, -1).copy()
    eng = BatchedDDMPC(n=n, m=m, p=p, L_=spec.L, N=u_d.shape[1], Q=spec.Q, R=spec.R, u_s=spec.u_s, y_s=spec.y_s,
                       batch=B, controller_type=L.ROBUST if spec.robust else L.NOMINAL,
                       slack_type=L.SLACK_CONVEX if spec.slack == "convex" else L.SLACK_NONE,
                       eps_max=spec.eps_max, lamb_alpha=spec.lamb_alpha, lamb_sigma=spec.lamb_sigma, c=spec.c,
                       use_terminal_constraint=spec.tec)
    print(tag, "kernel", eng.kernel_name(), "cost model", eng.cost_model(), flush=True)
    eng.set_data(u_d, y_d)
    t = time.time(); u, cost, st, it = eng.solve(up, yp); dt = time.time() - t
    worst_u = worst_c = 0.0
    for b in range(B):
        sol = orc.solve_fullspace(spec, u_d[b], y_d[b], up[b], yp[b])
        eu = np.max(np.abs(u[b] - sol.optimal_u)) / max(np.max(np.abs(sol.optimal_u)), 1e-300)
        ec = abs(cost[b] - sol.cost) / max(abs(sol.cost), 1e-9)
        worst_u, worst_c = max(worst_u, eu), max(worst_c, ec)
        if b < 3:
            print("  b", b, "status", st[b], "iters", it[b], "u0", u[b, :2], "ref", sol.optimal_u[:2], "cost", cost[b], sol.cost, "rel", eu, ec, "oracle iters", sol.iters)
    print(tag, "B", B, "worst rel err u %.3e cost %.3e" % (worst_u, worst_c), "status set", set(st.tolist()), "time %.3fs" % dt, flush=True)
    if spec.robust:
        a = eng.get_solution("alpha"); sg = eng.get_solution("sigma"); ub = eng.get_solution("ubar"); yb = eng.get_solution("ybar")
        sol = orc.solve_fullspace(spec, u_d[0], y_d[0], up[0], yp[0])
        print("  alpha err %.2e sigma err %.2e ubar err %.2e ybar err %.2e" % (np.max(np.abs(a[0]-sol.alpha)), np.max(np.abs(sg[0]-sol.sigma)), np.max(np.abs(ub[0]-sol.ubar)), np.max(np.abs(yb[0]-sol.ybar))))
    eng.close()

if __name__ == "__main__":
    lib = L.load(); print("devices", lib.ddmpc_device_count())
    run("robust/none", 8)
    run("robust/convex", 8, slack_var_constraint_type=1)
    run("nominal", 4, controller_type=0)
    run("ucon", 4, tec=False)
    run("small L=10 N=120", 4, L=10, N=120)
