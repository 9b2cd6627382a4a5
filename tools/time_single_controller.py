"""Dev timing: one DirectDataDrivenMPCController (batch 1) driven like the reference's closed loop --
construction (cold solve) and per-step latency of update_and_solve / get / store, warm and cold."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from direct_data_driven_mpc_amd.direct_data_driven_mpc_controller import (DataDrivenMPCType, DirectDataDrivenMPCController,
                                                                           SlackVarConstraintTypes)
from direct_data_driven_mpc_amd.harness import FOUR_TANK as P, controller_params, generate_batch, simulate_batch

cfg = controller_params()
d = generate_batch([0])
for slack in (SlackVarConstraintTypes.NONE, SlackVarConstraintTypes.CONVEX):
    for warm in (True, False):
        t0 = time.perf_counter()
        c = DirectDataDrivenMPCController(n=4, m=2, p=2, u_d=d["u_d"][0], y_d=d["y_d"][0], L=30, Q=cfg["Q"] * np.eye(60),
                                          R=cfg["R"] * np.eye(60), u_s=cfg["u_s"].reshape(-1, 1), y_s=cfg["y_s"].reshape(-1, 1),
                                          eps_max=cfg["eps_max"], lamb_alpha=cfg["lamb_alpha"], lamb_sigma=cfg["lamb_sigma"],
                                          c=cfg["c"], slack_var_constraint_type=slack, controller_type=DataDrivenMPCType.ROBUST)
        c.use_warm_steps = warm
        t_create = time.perf_counter() - t0
        x = d["x_end"][0].copy()
        rng = np.random.default_rng(1)
        ts = []
        for k in range(300):
            t1 = time.perf_counter()
            c.update_and_solve_data_driven_mpc()
            u = c.get_optimal_control_input_at_step(0)
            ts.append(time.perf_counter() - t1)
            y = P["C"] @ x + P["D"] @ u + 0.002 * rng.uniform(-1, 1, 2)
            x = P["A"] @ x + P["B"] @ u
            c.store_input_output_measurement(u.reshape(-1, 1), y.reshape(-1, 1))
        ts = np.array(ts[20:]) * 1e6
        print("slack %-6s warm=%-5s create %.1f ms; per step median %.0f us (p10 %.0f, p90 %.0f); y_end %s" % (
            slack.name, warm, t_create * 1e3, np.median(ts), np.percentile(ts, 10), np.percentile(ts, 90), y))
