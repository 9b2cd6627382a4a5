"""Batched counterpart of the reference's paper-reproduction script
(examples/robust_data_driven_mpc_reproduction.py:126-295): the three robust schemes of Berberich et al.,
Sec. V -- 1-step with terminal equality constraints (TEC), n-step TEC, 1-step without them (UCON) -- run
from the same data for `--batch` noise realisations at once, whole closed loops on the device.
Instance i uses seed `seed + i`; with `--seed 4 --batch 1` instance 0 is the run behind the reference's figure
(first inputs [8.6605, 8.5332] / [7.8963, 9.2947], UCON leaving |u| < 15 around step 384).  No plots: the
closed-loop data can be written to an .npz.

    python examples/batched_robust_reproduction.py --batch 1024 --seed 4
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
                                                plant_from_yaml, reproduction_start)

CFG = os.path.join(ROOT, "examples", "config")


def main():
    ap = argparse.ArgumentParser(description="Batched reproduction of the robust Data-Driven MPC example (MI355X)")
    ap.add_argument("--model_config_path", default=os.path.join(CFG, "models", "four_tank_system_params.yaml"))
    ap.add_argument("--model_key_value", default="FourTankSystem")
    ap.add_argument("--controller_config_path",
                    default=os.path.join(CFG, "controllers", "data_driven_mpc_example_params.yaml"))
    ap.add_argument("--controller_key_value", default="data_driven_mpc_params")
    ap.add_argument("--t_sim", type=int, default=600)
    ap.add_argument("--seed", type=int, default=4)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--y_0", type=float, nargs="+", default=[0.4, 0.4], help="output the plant starts from")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--out", default=None)
    ap.add_argument("--plot", default=None, help="write a PNG comparing the three schemes")
    a = ap.parse_args()

    plant = plant_from_yaml(a.model_config_path, a.model_key_value)
    m, p = plant["B"].shape[1], plant["C"].shape[0]
    cfg = controller_params_from_yaml(a.controller_config_path, a.controller_key_value, m=m, p=p,
                                      overrides=dict(controller_type=1, slack_var_constraint_type=0))
    n, B = cfg["n"], a.batch
    data = generate_batch(range(a.seed, a.seed + B), N=cfg["N"], plant=plant, u_range=cfg["u_range"])
    x_start, U_n, Y_n = reproduction_start(plant, data["rngs"], a.y_0, cfg["u_s"], n)
    n_steps = a.t_sim + 1 - n            # robust_data_driven_mpc_reproduction.py: n_steps = t_sim + 1, loops run n_steps - n
    results = {}
    for tag, tec, step in (("TEC 1-step", True, 1), ("TEC n-step", True, n), ("UCON 1-step", False, 1)):
        w = np.stack([plant["eps_max"] * rng.uniform(-1.0, 1.0, (n_steps, p)) for rng in data["rngs"]])   # per controller, in order
        eng = BatchedDDMPC(n=n, m=m, p=p, L_=cfg["L"], N=cfg["N"], Q=cfg["Q"], R=cfg["R"], u_s=cfg["u_s"], y_s=cfg["y_s"],
                           batch=B, controller_type=L.ROBUST, slack_type=L.SLACK_NONE, eps_max=cfg["eps_max"],
                           lamb_alpha=cfg["lamb_alpha"], lamb_sigma=cfg["lamb_sigma"], c=cfg["c"],
                           use_terminal_constraint=tec, device=a.device)
        eng.set_data(data["u_d"], data["y_d"])
        t0 = time.perf_counter()
        u_sys, y_sys, status, *_ = eng.closed_loop(plant["A"], plant["B"], plant["C"], plant["D"], x_start, U_n, Y_n, w,
                                                   n_mpc_step=step)
        dt = time.perf_counter() - t0
        eng.close()
        big = np.max(np.abs(u_sys), axis=2) > 15.0
        first_big = np.where(big.any(axis=1), big.argmax(axis=1) + n, -1)
        err = np.abs(y_sys[:, -1, :] - cfg["y_s"])
        print(f"{tag:12s}: {B} loops x {n_steps} steps in {dt * 1e3:7.1f} ms; first input of instance 0 {u_sys[0, 0]}; "
              f"final |y - y_s| median {np.median(err, axis=0)}; |u| > 15 in {int((first_big >= 0).sum())} instance(s)"
              + (f", instance 0 at step {int(first_big[0])}" if first_big[0] >= 0 else ""))
        results[tag] = (u_sys, y_sys, status)
    if a.plot:
        from _plot import plot_closed_loops
        plot_closed_loops(a.plot, {k: (r[0], r[1]) for k, r in results.items()}, cfg["u_s"], cfg["y_s"], t0=n,
                          title=f"robust Data-Driven MPC, {B} noise realisations from seed {a.seed}")
    if a.out:
        np.savez_compressed(a.out, **{k.replace(" ", "_") + "_" + nm: v for k, r in results.items()
                                      for nm, v in zip(("u", "y", "status"), r)})


if __name__ == "__main__":
    main()
