"""CPU tests against tests/golden/reference_loops.npz -- outputs of the REFERENCE's own driver code
(controller_creation.py, controller_operation.py, paper_reproduction.py, LTISystemModel) run in the build
container with a script-local controller whose solve is the oracle (tests/golden/make_golden_loops.py).
They pin this repo's restatements of parameter derivation, RNG/call order, loop order, FIFO updates and
the printed step line to the reference's code.  No GPU."""
import io
import os
import re
from contextlib import redirect_stdout

import numpy as np
import pytest

from direct_data_driven_mpc_amd import harness
from oracle import ddmpc_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, "examples", "config")


@pytest.fixture(scope="module")
def loops():
    return np.load(os.path.join(ROOT, "tests", "golden", "reference_loops.npz"))


def test_parameter_derivation_equals_the_references(loops):
    # get_data_driven_mpc_controller_params(reference YAML) vs harness.controller_params_from_yaml(own YAML)
    plant = harness.plant_from_yaml(os.path.join(CFG, "models", "four_tank_system_params.yaml"))
    cfg = harness.controller_params_from_yaml(os.path.join(CFG, "controllers", "data_driven_mpc_example_params.yaml"),
                                              m=plant["B"].shape[1], p=plant["C"].shape[0])
    for k in ("N", "n", "L", "eps_max", "lamb_alpha", "lamb_sigma", "c", "n_mpc_step"):
        assert float(cfg[k]) == float(loops["params_" + k][0]), k
    assert tuple(float(v) for v in cfg["u_range"]) == tuple(loops["params_u_range"])
    q, qr, qc = loops["params_Q_scalar_shape"]; r, rr, rc = loops["params_R_scalar_shape"]
    assert loops["params_Q_is_scaled_identity"][0] == 1
    assert (cfg["Q"], cfg["R"]) == (q, r) and (qr, qc) == (cfg["p"] * cfg["L"],) * 2 and (rr, rc) == (cfg["m"] * cfg["L"],) * 2
    assert np.array_equal(cfg["u_s"], loops["params_u_s"]) and np.array_equal(cfg["y_s"], loops["params_y_s"])
    assert [("ROBUST" if cfg["robust"] else "NOMINAL"), cfg["slack"].upper()] == list(loops["params_types"])
    # the class-level enums carry the same names the reference's maps produce
    from direct_data_driven_mpc_amd.direct_data_driven_mpc_controller import DataDrivenMPCType, SlackVarConstraintTypes
    assert DataDrivenMPCType[loops["params_types"][0]] is DataDrivenMPCType.ROBUST
    assert SlackVarConstraintTypes[loops["params_types"][1]] is SlackVarConstraintTypes.NONE


@pytest.mark.parametrize("tag,seed", [("ex_robust_s0", 0), ("ex_robust_s4", 4), ("ex_nominal_s0", 0)])
def test_data_generation_equals_the_reference_run(loops, tag, seed):
    d = harness.generate_batch([seed])
    assert np.array_equal(d["x_0"][0], loops[tag + "_x_0"])
    assert np.array_equal(d["u_d"][0], loops[tag + "_u_d"])
    assert np.array_equal(d["y_d"][0], loops[tag + "_y_d"])


class _OracleDriven:
    """Per-step methods of the reference class on the CPU oracle (test-local)."""

    def __init__(self, spec, u_d, y_d, n_mpc_step):
        self.spec, self.u_d, self.y_d, self.n_mpc_step = spec, u_d, y_d, n_mpc_step
        self.u_s, self.y_s = spec.u_s.reshape(-1, 1), spec.y_s.reshape(-1, 1)
        self.u_past, self.y_past = u_d[-spec.n:].reshape(-1, 1), y_d[-spec.n:].reshape(-1, 1)

    def update_and_solve_data_driven_mpc(self):
        self.sol = orc.solve_fullspace(self.spec, self.u_d, self.y_d, self.u_past, self.y_past)

    def get_optimal_control_input_at_step(self, n_step=0):
        return self.sol.optimal_u[n_step * self.spec.m:(n_step + 1) * self.spec.m]

    def get_optimal_cost_value(self):
        return self.sol.cost

    def store_input_output_measurement(self, u_current, y_current):
        self.u_past = np.vstack([self.u_past[self.spec.m:], u_current])
        self.y_past = np.vstack([self.y_past[self.spec.p:], y_current])


@pytest.mark.parametrize("tag,seed,kw,step", [("ex_robust_s0", 0, {}, 4), ("ex_robust_s4", 4, {}, 4),
                                              ("ex_convex1_s0", 0, dict(slack_var_constraint_type=1), 1),
                                              ("ex_nominal_s0", 0, dict(controller_type=0), 4)])
def test_loop_driver_and_oracle_loop_equal_the_reference_run(loops, tag, seed, kw, step):
    # harness.simulate_control_loop (what the cfg-1 GPU test drives the class with) and oracle.closed_loop
    # (what the device closed loop is checked against) both reproduce the reference's own loop code
    u_ref, y_ref = loops[tag + "_u_sys"], loops[tag + "_y_sys"]
    n_steps = u_ref.shape[0]
    spec = orc.spec_from_params(**kw)
    d = harness.generate_batch([seed])
    ctrl = _OracleDriven(spec, d["u_d"][0], d["y_d"][0], step)
    ctrl.update_and_solve_data_driven_mpc()                     # the constructor's solve (controller.py:239-240)
    buf = io.StringIO()
    with redirect_stdout(buf):
        u_sys, y_sys, _ = harness.simulate_control_loop(harness.FOUR_TANK, d["x_end"][0], ctrl, n_steps, d["rngs"][0], verbose=2)
    assert np.array_equal(u_sys, u_ref) and np.array_equal(y_sys, y_ref)
    assert buf.getvalue().splitlines() == list(loops[tag + "_lines"])
    inst = orc.generate_instance(seed)
    w = inst["plant"].eps_max * inst["rng"].uniform(-1.0, 1.0, (n_steps, 2))
    u2, y2 = orc.closed_loop(spec, inst["u_d"], inst["y_d"], inst["plant"], w, n_mpc_step=step)
    assert np.array_equal(u2, u_ref) and np.array_equal(y2, y_ref)


def test_step_line_format(loops):
    pat = re.compile(r"^    Time step: +\d+ - MPC cost value: +-?\d+\.\d{4} - Error: u_1e = +-?\d\.\d{3}, u_2e = +-?\d\.\d{3}, "
                     r"y_1e = +-?\d\.\d{3}, y_2e = +-?\d\.\d{3}$")
    for ln in loops["ex_robust_s0_lines"]:
        assert pat.match(ln), ln
    assert harness.step_report_line(400, 0.12474, [1, 1], [0.65, 0.77], [0.934, 0.917], [0.652, 0.77]) == \
        "    Time step:  400 - MPC cost value:   0.1247 - Error: u_1e =  0.066, u_2e =  0.083, y_1e = -0.002, y_2e =  0.000"


@pytest.mark.parametrize("seed", [0, 4])
def test_reproduction_start_and_loops_equal_the_reference_run(loops, seed):
    # examples/robust_data_driven_mpc_reproduction.py:126-290 run from the reference: equilibrium start, n warm-up
    # steps at u_s, then the three controllers one after the other, each loop drawing its own (n_loop, p) noise
    pre = "rep_s%d_" % seed
    d = harness.generate_batch([seed])
    assert np.array_equal(d["u_d"][0], loops[pre + "u_d"]) and np.array_equal(d["y_d"][0], loops[pre + "y_d"])
    cfg = harness.controller_params()
    n = cfg["n"]
    x_start, U_n, Y_n = harness.reproduction_start(harness.FOUR_TANK, d["rngs"], [0.4, 0.4], cfg["u_s"], n)
    assert np.array_equal(U_n[0].reshape(-1, 2), loops[pre + "U_n"])
    assert np.max(np.abs(Y_n[0].reshape(-1, 2) - loops[pre + "Y_n"])) < 1e-15
    plant = orc.Plant(**orc.FOUR_TANK)
    n_loop = 600 + 1 - n                                  # n_steps = t_sim + 1; the loops run n_steps - n
    for tag, tec, step in (("tec", True, 1), ("tec_nstep", True, n), ("ucon", False, 1)):
        assert list(loops[pre + tag + "_cfg"]) == [step, int(tec)]
        u_ref, y_ref = loops[pre + tag + "_u"], loops[pre + tag + "_y"]
        assert u_ref.shape == (n_loop, 2) and y_ref.shape == (n_loop, 2)
        w = plant.eps_max * d["rngs"][0].uniform(-1.0, 1.0, (n_loop, 2))        # drawn for every controller, in order
        if tag != "tec_nstep" and seed != 4:
            continue                                      # a 1-step loop is 597 oracle solves: seed 4 only
        spec = orc.spec_from_params(tec=tec)
        plant.x = x_start[0].copy()
        u2, y2 = orc.closed_loop(spec, d["u_d"][0], d["y_d"][0], plant, w, n_mpc_step=step, u_past=U_n[0], y_past=Y_n[0])
        assert np.max(np.abs(u2 - u_ref)) / np.max(np.abs(u_ref)) < 1e-9
        assert np.max(np.abs(y2 - y_ref)) < 1e-11
