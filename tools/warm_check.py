"""GPU check + timing of the warm path (ddmpc_prepare / ddmpc_step / fused closed loop) -- dev tool."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from direct_data_driven_mpc_amd import _lib as L
from direct_data_driven_mpc_amd.engine import BatchedDDMPC
from direct_data_driven_mpc_amd import harness
from oracle import ddmpc_oracle as orc


def make(B, **kw):
    spec = orc.spec_from_params(**kw)
    insts = [orc.generate_instance(s, N=kw.get("N", 400)) for s in range(B)]
    u_d = np.stack([i["u_d"] for i in insts]); y_d = np.stack([i["y_d"] for i in insts])
    n, m, p = spec.n, spec.m, spec.p
    eng = BatchedDDMPC(n=n, m=m, p=p, L_=spec.L, N=u_d.shape[1], Q=spec.Q, R=spec.R, u_s=spec.u_s, y_s=spec.y_s,
                       batch=B, controller_type=L.ROBUST if spec.robust else L.NOMINAL,
                       slack_type=L.SLACK_CONVEX if spec.slack == "convex" else L.SLACK_NONE,
                       eps_max=spec.eps_max, lamb_alpha=spec.lamb_alpha, lamb_sigma=spec.lamb_sigma, c=spec.c,
                       use_terminal_constraint=spec.tec)
    eng.set_data(u_d, y_d)
    return spec, eng, u_d, y_d, insts


def check(tag, B, **kw):
    spec, eng, u_d, y_d, insts = make(B, **kw)
    n = spec.n
    rng = np.random.default_rng(7)
    worst = 0.0
    for trial in range(3):
        if trial == 0:
            up = u_d[:, -n:, :].reshape(B, -1).copy(); yp = y_d[:, -n:, :].reshape(B, -1).copy()
        else:
            up = rng.uniform(-1, 1, (B, n * spec.m)); yp = rng.uniform(0, 1, (B, n * spec.p))
        uc, cc, sc, ic = eng.solve(up, yp)
        uw, cw, sw, iw = eng.step(up, yp)
        eu = np.max(np.abs(uw - uc)) / np.max(np.abs(uc)); ec = np.max(np.abs(cw - cc) / np.abs(cc))
        sol = orc.solve_fullspace(spec, u_d[0], y_d[0], up[0], yp[0])
        eo = np.max(np.abs(uw[0] - sol.optimal_u)) / np.max(np.abs(sol.optimal_u))
        worst = max(worst, eu, ec, eo)
        print("  trial", trial, "warm vs cold u %.2e cost %.2e | warm vs oracle u %.2e cost %.2e status" % (
            eu, ec, eo, abs(cw[0] - sol.cost) / abs(sol.cost)), set(sw.tolist()), set(iw.tolist()))
    if spec.robust:
        sg = eng.get_solution("sigma"); a = eng.get_solution("alpha")
        print("  get_solution after warm step: sigma err %.2e alpha err %.2e" % (
            np.max(np.abs(sg[0] - sol.sigma)), np.max(np.abs(a[0] - sol.alpha))))
    print(tag, "worst %.3e" % worst, flush=True)
    eng.close()


def closed_loop(tag, B, n_mpc_step, t_sim=100, **kw):
    spec, eng, u_d, y_d, insts = make(B, **kw)
    n = spec.n
    P = orc.FOUR_TANK
    x0 = np.stack([i["plant"].x for i in insts])
    up = u_d[:, -n:, :].reshape(B, -1).copy(); yp = y_d[:, -n:, :].reshape(B, -1).copy()
    w = np.stack([np.random.default_rng(1000 + b).uniform(-1, 1, (t_sim + 1, spec.p)) * 0.002 for b in range(B)])
    outs = {}
    for path in ("cold", "warm"):
        eng.set_closed_loop_path(path)
        t = time.time()
        outs[path] = eng.closed_loop(P["A"], P["B"], P["C"], P["D"], x0, up, yp, w, n_mpc_step=n_mpc_step)
        print("  ", path, "%.3f s" % (time.time() - t))
    for name, a, b in zip(("u_sys", "y_sys", "status", "x", "up", "yp"), outs["cold"], outs["warm"]):
        print("   %s max abs diff %.3e" % (name, np.max(np.abs(np.asarray(a, dtype=float) - np.asarray(b, dtype=float)))))
    print(tag, "y_end", outs["warm"][1][0, -1], flush=True)
    eng.close()


def timing(B=4096):
    import torch
    d = harness.generate_batch(range(B), N=400)
    u_d, y_d = d["u_d"], d["y_d"]
    n, m, p = 4, 2, 2
    eng = BatchedDDMPC(n=n, m=m, p=p, L_=30, N=400, Q=3.0, R=1e-4, u_s=[1, 1], y_s=[0.65, 0.77], batch=B,
                       controller_type=L.ROBUST, slack_type=L.SLACK_NONE, eps_max=0.002, lamb_alpha=50.0,
                       lamb_sigma=1000.0, c=1.0)
    dev = torch.device("cuda:0")
    tu = torch.from_numpy(u_d).to(dev); ty = torch.from_numpy(y_d).to(dev)
    up = tu[:, -n:, :].reshape(B, -1).contiguous(); yp = ty[:, -n:, :].reshape(B, -1).contiguous()
    eng.set_data(tu, ty)
    torch.cuda.synchronize(); t = time.time(); eng.prepare(); torch.cuda.synchronize()
    print("prepare (factor export + gain) %.3f ms for B=%d" % ((time.time() - t) * 1e3, B))
    out = eng.step(up, yp)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    K = 200
    e0.record()
    for _ in range(K):
        eng.step(up, yp, *out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / K
    nf = n * (m + p); r = (m + p) * (30 + n)
    bytes_step = 8.0 * ((nf + 1) * r + nf + 30 * m + 1) + 8
    print("warm step %.4f ms/step  %.3e steps/s  algorithmic %.1f GB/s (%.1f KB/step)" % (
        ms, B / ms * 1e3, bytes_step * B / ms / 1e6, bytes_step / 1024))
    eng.close()


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "time"):
        import torch
        torch.cuda.init()                 # torch first: it must see the device before the library's runtime does
    lib = L.load(); print("devices", lib.ddmpc_device_count())
    if which in ("all", "check"):
        check("robust/none", 8)
        check("nominal", 4, controller_type=0)
        check("ucon", 4, tec=False)
        check("small L=10 N=120", 4, L=10, N=120)
    if which in ("all", "loop"):
        closed_loop("loop 1-step", 8, 1)
        closed_loop("loop n-step", 8, 4)
        closed_loop("loop ucon", 4, 1, tec=False)
    if which in ("all", "time"):
        timing()
        timing(32768)
