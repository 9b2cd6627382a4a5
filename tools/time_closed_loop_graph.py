"""Dev timing: per-step closed-loop paths with and without HIP-graph replay."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from direct_data_driven_mpc_amd import _lib as L
from direct_data_driven_mpc_amd.engine import BatchedDDMPC
from direct_data_driven_mpc_amd.harness import FOUR_TANK as P, controller_params, generate_batch

B, n_steps = 4096, 401
d = generate_batch(range(B))
w = 0.002 * np.random.default_rng(1).uniform(-1, 1, (B, n_steps, 2))
up = d["u_d"][:, -4:, :].reshape(B, -1); yp = d["y_d"][:, -4:, :].reshape(B, -1)
for slack, path, step in ((L.SLACK_CONVEX, "auto", 1), (L.SLACK_CONVEX, "auto", 4), (L.SLACK_NONE, "cold", 4)):
    cfg = controller_params()
    eng = BatchedDDMPC(n=4, m=2, p=2, L_=30, N=400, Q=cfg["Q"], R=cfg["R"], u_s=cfg["u_s"], y_s=cfg["y_s"], batch=B,
                       controller_type=L.ROBUST, slack_type=slack, eps_max=cfg["eps_max"], lamb_alpha=cfg["lamb_alpha"],
                       lamb_sigma=cfg["lamb_sigma"], c=cfg["c"])
    eng.set_data(d["u_d"], d["y_d"]); eng.set_closed_loop_path(path)
    for graph in (True, False, True, False):
        eng.set_closed_loop_graph(graph)
        t = time.perf_counter()
        out = eng.closed_loop(P["A"], P["B"], P["C"], P["D"], d["x_end"], up, yp, w, n_mpc_step=step)
        print("slack %d path %s n_mpc_step %d graph %-5s: %.1f ms" % (slack, path, step, graph, (time.perf_counter() - t) * 1e3), flush=True)
    eng.close()
