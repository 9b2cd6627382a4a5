"""Batched Direct Data-Driven MPC example on an MI355X.

The flow of the reference's example script (examples/direct_data_driven_mpc_example.py:169-425) --
load the model and controller YAML files, randomise the initial state, generate a persistently
exciting trajectory, create the controller, close the loop for `t_sim` steps -- for `--batch`
independent noise realisations (seeds `seed .. seed+batch-1`) at once, with the whole control loop
on the device (ddmpc_closed_loop).  Instance `i` is the problem the reference example builds with
`--seed <seed+i>`.  No plots / animation: the closed-loop data are optionally written to an .npz.

    python examples/batched_data_driven_mpc_example.py --batch 4096 --t_sim 400 --verbose 1
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples"))

from direct_data_driven_mpc_amd import _lib as L                                  # noqa: E402
from direct_data_driven_mpc_amd.engine import BatchedDDMPC                        # noqa: E402
from direct_data_driven_mpc_amd.harness import (controller_params_from_yaml, generate_batch,   # noqa: E402
                                                plant_from_yaml, step_report_line)

CFG = os.path.join(ROOT, "examples", "config")


def parse_args():
    ap = argparse.ArgumentParser(description="Batched Direct Data-Driven MPC example (MI355X)")
    ap.add_argument("--model_config_path", default=os.path.join(CFG, "models", "four_tank_system_params.yaml"))
    ap.add_argument("--model_key_value", default="FourTankSystem")
    ap.add_argument("--controller_config_path",
                    default=os.path.join(CFG, "controllers", "data_driven_mpc_example_params.yaml"))
    ap.add_argument("--controller_key_value", default="data_driven_mpc_params")
    ap.add_argument("--n_mpc_step", type=int, default=None, help="n-step scheme: inputs applied per solve")
    ap.add_argument("--controller_type", choices=["Nominal", "Robust"], default=None)
    ap.add_argument("--slack_var_const_type", choices=["None", "Convex", "NonConvex"], default=None)
    ap.add_argument("--t_sim", type=int, default=400)
    ap.add_argument("--seed", type=int, default=0, help="seed of instance 0; instance i uses seed+i")
    ap.add_argument("--batch", type=int, default=1024, help="number of independent controller instances")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--out", default=None, help="write u_sys / y_sys / status of all instances to this .npz")
    ap.add_argument("--plot", default=None, help="write a PNG of the closed-loop inputs/outputs (median, band, instance 0)")
    ap.add_argument("--verbose", type=int, choices=[0, 1, 2], default=1)
    return ap.parse_args()


def main():
    a = parse_args()
    plant = plant_from_yaml(a.model_config_path, a.model_key_value)
    m, p = plant["B"].shape[1], plant["C"].shape[0]
    over = {}
    if a.controller_type is not None:
        over["controller_type"] = {"Nominal": 0, "Robust": 1}[a.controller_type]
    if a.slack_var_const_type is not None:
        over["slack_var_constraint_type"] = {"None": 0, "Convex": 1, "NonConvex": 2}[a.slack_var_const_type]
    cfg = controller_params_from_yaml(a.controller_config_path, a.controller_key_value, m=m, p=p, overrides=over)
    n_mpc_step = a.n_mpc_step if a.n_mpc_step is not None else cfg["n_mpc_step"]   # controller_creation.py: n_mpc_step = n
    n, Lh, N, B = cfg["n"], cfg["L"], cfg["N"], a.batch
    if cfg["slack"] == "non_convex":
        raise NotImplementedError("Robust Data-Driven MPC with a non-convex constraint for the slack variable "
                                  "is not currently implemented.")            # controller.py:664-670
    if a.verbose:
        print(f"Data-Driven MPC: {'robust' if cfg['robust'] else 'nominal'} scheme, slack {cfg['slack']}, "
              f"n={n} L={Lh} N={N}, n_mpc_step={n_mpc_step}, batch={B}, seeds {a.seed}..{a.seed + B - 1}")

    t0 = time.perf_counter()
    data = generate_batch(range(a.seed, a.seed + B), N=N, plant=plant, u_range=cfg["u_range"])
    n_steps = a.t_sim + 1                                                        # controller_operation.py:263
    w = np.stack([plant["eps_max"] * rng.uniform(-1.0, 1.0, (n_steps, p)) for rng in data["rngs"]])
    t_gen = time.perf_counter() - t0

    eng = BatchedDDMPC(n=n, m=m, p=p, L_=Lh, N=N, Q=cfg["Q"], R=cfg["R"], u_s=cfg["u_s"], y_s=cfg["y_s"], batch=B,
                       controller_type=L.ROBUST if cfg["robust"] else L.NOMINAL,
                       slack_type=L.SLACK_CONVEX if cfg["slack"] == "convex" else L.SLACK_NONE,
                       eps_max=cfg["eps_max"], lamb_alpha=cfg["lamb_alpha"], lamb_sigma=cfg["lamb_sigma"], c=cfg["c"],
                       use_terminal_constraint=cfg["tec"], device=a.device)
    ok, rank = eng.persistent_excitation_guard(data["u_d"])                      # controller.py:275-296
    if not np.all(ok):
        bad = int(np.nonzero(~ok)[0][0])
        raise ValueError(f"Initial input trajectory data is not persistently exciting of order (L + 2 * n) "
                         f"(instance {bad}: rank {int(rank[bad])}).")
    eng.set_data(data["u_d"], data["y_d"])
    up = data["u_d"][:, -n:, :].reshape(B, -1)                                   # controller.py:184-185
    yp = data["y_d"][:, -n:, :].reshape(B, -1)
    u0, cost0, status0, _ = eng.solve(up, yp)                                    # the construction-time solve
    if np.any(status0 > 1):
        raise ValueError("Failed to get the optimal control input: the first solve is not optimal for "
                         f"{int(np.count_nonzero(status0 > 1))} instance(s)")
    t1 = time.perf_counter()
    u_sys, y_sys, status, x_end, up_end, yp_end = eng.closed_loop(plant["A"], plant["B"], plant["C"], plant["D"],
                                                                  data["x_end"], up, yp, w, n_mpc_step=n_mpc_step)
    t_loop = time.perf_counter() - t1

    if a.verbose > 1:                                                            # per-step report of instance 0
        with BatchedDDMPC(n=n, m=m, p=p, L_=Lh, N=N, Q=cfg["Q"], R=cfg["R"], u_s=cfg["u_s"], y_s=cfg["y_s"], batch=1,
                          controller_type=L.ROBUST if cfg["robust"] else L.NOMINAL,
                          slack_type=L.SLACK_CONVEX if cfg["slack"] == "convex" else L.SLACK_NONE,
                          eps_max=cfg["eps_max"], lamb_alpha=cfg["lamb_alpha"], lamb_sigma=cfg["lamb_sigma"],
                          c=cfg["c"], use_terminal_constraint=cfg["tec"], device=a.device) as one:
            one.set_data(data["u_d"][:1], data["y_d"][:1])
            U = np.concatenate([data["u_d"][0, -n:], u_sys[0]]); Y = np.concatenate([data["y_d"][0, -n:], y_sys[0]])
            for t in range(0, n_steps, n_mpc_step):                              # one line per solve, controller_operation.py:310-329
                _, cst, _, _ = one.step(U[t:t + n].reshape(1, -1), Y[t:t + n].reshape(1, -1))
                k = min(t + n_mpc_step, n_steps) - 1                             # errors of the last applied step
                print(step_report_line(t, float(cst[0]), cfg["u_s"], cfg["y_s"], u_sys[0, k], y_sys[0, k]))
    if a.verbose:
        n_bad = int(np.count_nonzero(status > 1))
        err_y = np.abs(y_sys[:, -1, :] - cfg["y_s"]); err_u = np.abs(u_sys[:, -1, :] - cfg["u_s"])
        solves = B * ((n_steps + n_mpc_step - 1) // n_mpc_step)
        print(f"data generation (host) {t_gen:.2f} s; closed loop of {B} controllers x {n_steps} steps "
              f"({solves} QP solves) in {t_loop * 1e3:.1f} ms incl. host<->device copies")
        print(f"non-optimal instances: {n_bad}; final |y - y_s| mean {err_y.mean(axis=0)}, max {err_y.max(axis=0)}; "
              f"final |u - u_s| mean {err_u.mean(axis=0)}")
        print(f"instance 0: y[-1] = {y_sys[0, -1]}, u[-1] = {u_sys[0, -1]}")
    if a.out:
        np.savez_compressed(a.out, u_sys=u_sys, y_sys=y_sys, status=status, u_s=cfg["u_s"], y_s=cfg["y_s"])
    if a.plot:
        from _plot import plot_closed_loops
        plot_closed_loops(a.plot, {"closed loop": (u_sys, y_sys)}, cfg["u_s"], cfg["y_s"],
                          title=f"{B} controllers, {'robust' if cfg['robust'] else 'nominal'} scheme, slack {cfg['slack']}")
    eng.close()


if __name__ == "__main__":
    main()
