"""Dev timing: ddmpc_closed_loop for 1 / 512 / 4096 controllers (1-step and 4-step schemes), host arrays in and out."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from direct_data_driven_mpc_amd import _lib as L
from direct_data_driven_mpc_amd.engine import BatchedDDMPC
from direct_data_driven_mpc_amd.harness import controller_params, generate_batch, FOUR_TANK as P
cfg = controller_params()
for B, n_steps, step in ((1, 596, 1), (512, 201, 4), (4096, 401, 4)):
    t0 = time.perf_counter(); d = generate_batch(range(B)); t1 = time.perf_counter()
    w = 0.002 * np.random.default_rng(1).uniform(-1, 1, (B, n_steps, 2))
    up = d["u_d"][:, -4:, :].reshape(B, -1); yp = d["y_d"][:, -4:, :].reshape(B, -1)
    eng = BatchedDDMPC(n=4, m=2, p=2, L_=30, N=400, Q=cfg["Q"], R=cfg["R"], u_s=cfg["u_s"], y_s=cfg["y_s"], batch=B,
                       eps_max=cfg["eps_max"], lamb_alpha=cfg["lamb_alpha"], lamb_sigma=cfg["lamb_sigma"], c=cfg["c"])
    eng.set_data(d["u_d"], d["y_d"]); t2 = time.perf_counter()
    out = eng.closed_loop(P["A"], P["B"], P["C"], P["D"], d["x_end"], up, yp, w, n_mpc_step=step); t3 = time.perf_counter()
    nsolve = -(-n_steps // step)
    print("B=%d steps=%d n_mpc_step=%d: gen %.2fs create+set %.2fs closed_loop %.3fs = %d solves -> %.3e solves/s; y_end[0]=%s status_ok=%d" % (
        B, n_steps, step, t1 - t0, t2 - t1, t3 - t2, nsolve * B, nsolve * B / (t3 - t2), out[1][0, -1], int((out[2] == 0).sum())), flush=True)
    eng.close()
