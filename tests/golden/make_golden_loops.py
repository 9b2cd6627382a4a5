"""Generate tests/golden/reference_loops.npz: outputs of the REFERENCE's own driver code, run in the build
container (the only place /root/reference exists):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_loops.py

What runs from the reference itself (imported from /root/reference, never copied):
  * utilities.controller.controller_creation.get_data_driven_mpc_controller_params / create_data_driven_mpc_controller
    on the reference's own controller YAML (parameter derivation, controller_creation.py:105-168,255-273);
  * utilities.controller.controller_operation.randomize_initial_system_state / generate_initial_input_output_data /
    simulate_n_input_output_measurements / simulate_data_driven_mpc_control_loop (RNG order, loop order,
    the printed per-step line, controller_operation.py:59-75,126-133,190-197,259-331);
  * utilities.reproduction.paper_reproduction (controller schemes, equilibrium start, the per-controller loops,
    paper_reproduction.py:43-59,65-90,91-150,151-202) in the sequence of
    examples/robust_data_driven_mpc_reproduction.py:126-290;
  * utilities.model_simulation.LTISystemModel on the reference's four-tank YAML.

What does NOT come from the reference: the QP solve.  The reference's controller module imports cvxpy, which is
not installed anywhere in this pipeline; the module `direct_data_driven_mpc.direct_data_driven_mpc_controller`
those files import is therefore provided HERE, script-locally, by a class with the reference constructor's
signature whose solve is oracle/ddmpc_oracle.py (never the product).  The fixtures thus pin call order, RNG
order, parameter derivation, FIFO updates and print format to the reference's code, with the oracle's QP optimum.
"""
import contextlib
import enum
import io
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)

from oracle import ddmpc_oracle as orc  # noqa: E402


# ---- script-local stand-in for the cvxpy-based controller module -------------------------------------------
class DataDrivenMPCType(enum.Enum):
    NOMINAL = 0,
    ROBUST = 1


class SlackVarConstraintTypes(enum.Enum):
    NON_CONVEX = 0,
    CONVEX = 1,
    NONE = 2


class OracleController:
    """Constructor signature and method surface the reference's driver code uses; solve = numpy full-space oracle."""

    def __init__(self, n, m, p, u_d, y_d, L, Q, R, u_s, y_s, eps_max=None, lamb_alpha=None, lamb_sigma=None, c=None,
                 slack_var_constraint_type=SlackVarConstraintTypes.CONVEX, controller_type=DataDrivenMPCType.NOMINAL,
                 n_mpc_step=1, use_terminal_constraint=True):
        self.n, self.m, self.p, self.L = n, m, p, L
        self.u_d, self.y_d, self.u_s, self.y_s, self.n_mpc_step = u_d, y_d, u_s, y_s, n_mpc_step
        robust = controller_type == DataDrivenMPCType.ROBUST
        slack = {SlackVarConstraintTypes.NONE: "none", SlackVarConstraintTypes.CONVEX: "convex"}[slack_var_constraint_type]
        self.spec = orc.QPSpec(n=n, m=m, p=p, L=L, Q=Q, R=R, u_s=np.asarray(u_s).reshape(-1), y_s=np.asarray(y_s).reshape(-1),
                               robust=robust, eps_max=eps_max, lamb_alpha=lamb_alpha, lamb_sigma=lamb_sigma, c=c,
                               slack=slack, tec=use_terminal_constraint)
        self.u_past = u_d[-n:].reshape(-1, 1)
        self.y_past = y_d[-n:].reshape(-1, 1)
        self.n_solves = 0
        self.update_and_solve_data_driven_mpc()

    def update_and_solve_data_driven_mpc(self):
        self.sol = orc.solve_fullspace(self.spec, self.u_d, self.y_d, self.u_past, self.y_past)
        assert self.sol.status == "optimal"
        self.n_solves += 1

    def get_optimal_control_input_at_step(self, n_step=0):
        return self.sol.optimal_u[n_step * self.m:(n_step + 1) * self.m]

    def get_optimal_cost_value(self):
        return self.sol.cost

    def store_input_output_measurement(self, u_current, y_current):
        self.u_past = np.vstack([self.u_past[self.m:], u_current])
        self.y_past = np.vstack([self.y_past[self.p:], y_current])

    def set_past_input_output_data(self, u_past, y_past):
        self.u_past, self.y_past = u_past, y_past


pkg = types.ModuleType("direct_data_driven_mpc")
pkg.__path__ = []
mod = types.ModuleType("direct_data_driven_mpc.direct_data_driven_mpc_controller")
mod.DirectDataDrivenMPCController = OracleController
mod.DataDrivenMPCType = DataDrivenMPCType
mod.SlackVarConstraintTypes = SlackVarConstraintTypes
sys.modules["direct_data_driven_mpc"] = pkg
sys.modules["direct_data_driven_mpc.direct_data_driven_mpc_controller"] = mod
sys.path.insert(1, REF)

import matplotlib  # noqa: E402
matplotlib.use("Agg")
from utilities.controller.controller_creation import (  # noqa: E402  (reference)
    create_data_driven_mpc_controller, get_data_driven_mpc_controller_params)
from utilities.controller.controller_operation import (  # noqa: E402  (reference)
    generate_initial_input_output_data, randomize_initial_system_state, simulate_data_driven_mpc_control_loop,
    simulate_n_input_output_measurements)
from utilities.model_simulation import LTISystemModel  # noqa: E402  (reference)
from utilities.reproduction.paper_reproduction import (  # noqa: E402  (reference)
    DataDrivenMPCScheme, create_data_driven_mpc_controllers_reproduction, get_equilibrium_state_from_output,
    simulate_data_driven_mpc_control_loops_reproduction)

MODEL_YAML = os.path.join(REF, "examples/config/models/four_tank_system_params.yaml")
CTRL_YAML = os.path.join(REF, "examples/config/controllers/data_driven_mpc_example_params.yaml")


def example_flow(seed, t_sim, controller_type=None, slack=None, n_mpc_step=None):
    """examples/direct_data_driven_mpc_example.py:169-330 (the part before the plots), verbose = 2 for the step lines."""
    model = LTISystemModel(config_file=MODEL_YAML, model_key_value="FourTankSystem")
    m, p = model.get_number_inputs(), model.get_number_outputs()
    cfg = get_data_driven_mpc_controller_params(config_file=CTRL_YAML, controller_key_value="data_driven_mpc_params", m=m, p=p)
    if n_mpc_step is not None:
        cfg["n_mpc_step"] = n_mpc_step
    if controller_type is not None:
        cfg["controller_type"] = controller_type
    if slack is not None:
        cfg["slack_var_constraint_type"] = slack
    rng = np.random.default_rng(seed=seed)
    x_0 = randomize_initial_system_state(system_model=model, controller_config=cfg, np_random=rng)
    model.set_state(state=x_0)
    u_d, y_d = generate_initial_input_output_data(system_model=model, controller_config=cfg, np_random=rng)
    ctrl = create_data_driven_mpc_controller(controller_config=cfg, u_d=u_d, y_d=y_d)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        u_sys, y_sys = simulate_data_driven_mpc_control_loop(system_model=model, data_driven_mpc_controller=ctrl,
                                                             n_steps=t_sim + 1, np_random=rng, verbose=2)
    lines = [ln for ln in buf.getvalue().splitlines() if "Time step" in ln]
    return dict(x_0=x_0, u_d=u_d, y_d=y_d, u_sys=u_sys, y_sys=y_sys, lines=np.array(lines), n_solves=np.array([ctrl.n_solves]))


def reproduction_flow(seed, t_sim):
    """examples/robust_data_driven_mpc_reproduction.py:126-290."""
    model = LTISystemModel(config_file=MODEL_YAML, model_key_value="FourTankSystem")
    m, p = model.get_number_inputs(), model.get_number_outputs()
    cfg = get_data_driven_mpc_controller_params(config_file=CTRL_YAML, controller_key_value="data_driven_mpc_params", m=m, p=p)
    rng = np.random.default_rng(seed=seed)
    x_0 = randomize_initial_system_state(system_model=model, controller_config=cfg, np_random=rng)
    model.set_state(state=x_0)
    u_d, y_d = generate_initial_input_output_data(system_model=model, controller_config=cfg, np_random=rng)
    schemes = [DataDrivenMPCScheme.TEC, DataDrivenMPCScheme.TEC_N_STEP, DataDrivenMPCScheme.UCON]
    ctrls = create_data_driven_mpc_controllers_reproduction(controller_config=cfg, u_d=u_d, y_d=y_d,
                                                            data_driven_mpc_controller_schemes=schemes)
    y_0 = [0.4, 0.4]                      # robust_data_driven_mpc_reproduction.py:77 (a plain list)
    xrep_0 = get_equilibrium_state_from_output(system_model=model, y_eq=y_0)
    model.set_state(xrep_0)
    U_n, Y_n = simulate_n_input_output_measurements(system_model=model, controller_config=cfg, np_random=rng)
    for c in ctrls:
        c.set_past_input_output_data(u_past=U_n.reshape(-1, 1), y_past=Y_n.reshape(-1, 1))
    n = cfg["n"]
    u_data, y_data = simulate_data_driven_mpc_control_loops_reproduction(
        system_model=model, data_driven_mpc_controllers=ctrls, n_steps=t_sim + 1 - n, np_random=rng, verbose=0)
    out = dict(xrep_0=np.asarray(xrep_0).reshape(-1), U_n=U_n, Y_n=Y_n, u_d=u_d, y_d=y_d)
    for tag, u, y, c in zip(("tec", "tec_nstep", "ucon"), u_data, y_data, ctrls):
        out[tag + "_u"], out[tag + "_y"] = u, y
        out[tag + "_cfg"] = np.array([c.n_mpc_step, int(c.spec.tec)])
    return out


def main():
    out = {}
    # (i) parameter derivation of the reference YAML
    cfg = get_data_driven_mpc_controller_params(config_file=CTRL_YAML, controller_key_value="data_driven_mpc_params", m=2, p=2)
    for k in ("N", "n", "L", "eps_max", "lamb_alpha", "lamb_sigma", "c", "n_mpc_step"):
        out["params_" + k] = np.array([cfg[k]], dtype=float)
    out["params_u_range"] = np.array(cfg["u_range"], dtype=float)
    out["params_Q_scalar_shape"] = np.array([cfg["Q"][0, 0], *cfg["Q"].shape], dtype=float)
    out["params_R_scalar_shape"] = np.array([cfg["R"][0, 0], *cfg["R"].shape], dtype=float)
    out["params_Q_is_scaled_identity"] = np.array([int(np.array_equal(cfg["Q"], cfg["Q"][0, 0] * np.eye(cfg["Q"].shape[0])))])
    out["params_u_s"], out["params_y_s"] = cfg["u_s"].reshape(-1), cfg["y_s"].reshape(-1)
    out["params_types"] = np.array([cfg["controller_type"].name, cfg["slack_var_constraint_type"].name])
    # (ii) the example flow: robust (YAML default, n-step scheme n_mpc_step = n) seeds 0 and 4; robust 1-step with the
    #      slack box, seed 0; nominal (BASELINE configs[0]) seed 0, t_sim = 400
    for tag, kw in (("ex_robust_s0", dict(seed=0, t_sim=400)), ("ex_robust_s4", dict(seed=4, t_sim=400)),
                    ("ex_convex1_s0", dict(seed=0, t_sim=60, slack=SlackVarConstraintTypes.CONVEX, n_mpc_step=1)),
                    ("ex_nominal_s0", dict(seed=0, t_sim=400, controller_type=DataDrivenMPCType.NOMINAL))):
        res = example_flow(**kw)
        for k, v in res.items():
            out[tag + "_" + k] = v
        print(tag, "solves", int(res["n_solves"][0]), "last line:", res["lines"][-1])
    # (iii) the paper-reproduction sequence, seeds 0 and 4 (the script's default), t_sim = 600
    for seed in (0, 4):
        res = reproduction_flow(seed, 600)
        for k, v in res.items():
            out["rep_s%d_%s" % (seed, k)] = v
        print("rep seed", seed, "first inputs", res["tec_u"][4], res["ucon_u"][4])
    path = os.path.join(HERE, "reference_loops.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, len(out), "arrays,", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
