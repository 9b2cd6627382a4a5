"""Dev tool: throughput of the other BASELINE configs / modes (not the headline bench line)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
from direct_data_driven_mpc_amd import _lib as L
from direct_data_driven_mpc_amd.engine import BatchedDDMPC
from direct_data_driven_mpc_amd.harness import controller_params, generate_batch
sys.path.insert(0, "tools")
from cfg5_time import large_flops, PEAK_TF      # algorithmic flop model of the global-workspace kernels

def run(tag, B, Lh, N, slack, gram=0, steps=10, host=False, pipelined=False):
    cfg = controller_params(dict(L=Lh, N=N, slack_var_constraint_type=slack))
    d = generate_batch(range(B), N=N)
    n, m, p = 4, 2, 2
    up = d["u_d"][:, -n:, :].reshape(B, -1).copy(); yp = d["y_d"][:, -n:, :].reshape(B, -1).copy()
    eng = BatchedDDMPC(n=n, m=m, p=p, L_=Lh, N=N, Q=cfg["Q"], R=cfg["R"], u_s=cfg["u_s"], y_s=cfg["y_s"], batch=B,
                       controller_type=L.ROBUST, slack_type=L.SLACK_CONVEX if slack else L.SLACK_NONE,
                       eps_max=cfg["eps_max"], lamb_alpha=cfg["lamb_alpha"], lamb_sigma=cfg["lamb_sigma"], c=cfg["c"], gram_mode=gram)
    dev = torch.device("cuda", 0)
    if host:
        eng.solve_from_host(d["u_d"], d["y_d"], up, yp)          # buffers allocated outside the timed region
        t0 = time.perf_counter()
        for _ in range(steps):
            if pipelined:
                eng.solve_from_host(d["u_d"], d["y_d"], up, yp)
            else:
                eng.set_data(d["u_d"], d["y_d"]); eng.solve(up, yp)
        dt = (time.perf_counter() - t0) / steps
    else:
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        ud, yd, upt, ypt = t(d["u_d"]), t(d["y_d"]), t(up), t(yp)
        eng.set_data(ud, yd)
        out = eng.solve(upt, ypt)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.solve(upt, ypt, *out)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        st = out[2].cpu().numpy(); it = out[3].cpu().numpy()
        tag += " status_ok=%d iters_mean=%.2f" % (int((st == 0).sum()), it.mean())
    f, b = eng.cost_model()
    tf = f * B / dt / 1e12
    print("%-70s kernel %-30s B=%6d  %.3f ms/step  %.3e solves/s  %.2f TFLOP/s(alg) = %.3f of the %.1f TF fp64-MFMA peak" % (
        tag, eng.kernel_name(), B, dt * 1e3, B / dt, tf, tf / PEAK_TF, PEAK_TF), flush=True)
    eng.close()

# host-pointer paths first: measured at the END of this process (after the 32,768-instance and <17,8> runs) the chunked path read
# 2.6-2.7 ms in rounds 3-4 while the stand-alone tool (tools/time_host_pipeline.py, tools/host_pipeline_order.py) has it at 1.37-1.41 ms,
# below the plain upload + solve, in every order of calls -- see DESIGN section 7
run("cfg2 PCIe-inclusive (host pointers: upload u_d,y_d + solve + download)", 4096, 30, 400, 0, host=True, steps=5)
run("cfg2 PCIe-inclusive, pipelined (ddmpc_solve_from_host: chunked upload overlapped with the solves)", 4096, 30, 400, 0, host=True, steps=5, pipelined=True)
run("cfg2 robust NONE structured", 4096, 30, 400, 0)
run("cfg2 robust NONE dense-MFMA Gram", 4096, 30, 400, 0, gram=1)
run("cfg2 robust CONVEX (slack box, active set)", 4096, 30, 400, 1)
run("cfg3 shard: robust NONE, 32768 per GPU", 32768, 30, 400, 0, steps=5)
run("cfg4 robust NONE L=60 N=1000", 1024, 60, 1000, 0)
run("cfg4 robust CONVEX L=60 N=1000", 1024, 60, 1000, 1)
run("(again, at the end of the process) cfg2 PCIe-inclusive", 4096, 30, 400, 0, host=True, steps=5)
run("(again, at the end of the process) cfg2 PCIe-inclusive, pipelined", 4096, 30, 400, 0, host=True, steps=5, pipelined=True)


def run_config5(B=512):
    """BASELINE configs[4]: nominal scheme, m = p = 8, n = 8, L = 30, N = 2000, exact data of a random stable plant
    (SURVEY section 8 proposal) -- the rank-revealing kernel with a global workspace."""
    rng = np.random.default_rng(0)
    ns = n = 8; m = p = 8; Lh = 30; N = 2000
    A = rng.normal(size=(ns, ns)); A *= 0.9 / max(abs(np.linalg.eigvals(A)))
    plant = dict(A=A, B=rng.normal(size=(ns, m)), C=rng.normal(size=(p, ns)), D=np.zeros((p, m)), eps_max=0.0)
    u_s = 0.1 * np.ones(m); y_s = (plant["C"] @ np.linalg.inv(np.eye(ns) - A) @ plant["B"]) @ u_s
    d = generate_batch(range(B), N=N, plant=plant)
    up = d["u_d"][:, -n:, :].reshape(B, -1).copy(); yp = d["y_d"][:, -n:, :].reshape(B, -1).copy()
    eng = BatchedDDMPC(n=n, m=m, p=p, L_=Lh, N=N, Q=3.0, R=1e-4, u_s=u_s, y_s=y_s, batch=B, controller_type=L.NOMINAL)
    dev = torch.device("cuda", 0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    ud, yd, upt, ypt = t(d["u_d"]), t(d["y_d"]), t(up), t(yp)
    eng.set_data(ud, yd)
    out = eng.solve(upt, ypt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        eng.solve(upt, ypt, *out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    st = out[2].cpu().numpy()
    tf = large_flops(m, p, n, Lh, N, False) * B / dt / 1e12
    print("%-70s kernel %-30s B=%6d  %.3f ms/step  %.3e solves/s  %.2f TFLOP/s(alg) = %.3f of the %.1f TF fp64-MFMA peak" % (
        "cfg5 nominal, m=p=8 n=8 L=30 N=2000, exact data status_ok=%d" % int((st == 0).sum()), eng.kernel_name(), B, dt * 1e3, B / dt,
        tf, tf / PEAK_TF, PEAK_TF), flush=True)
    eng.close()


run_config5()
