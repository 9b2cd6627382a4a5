"""CPU tests of the host-side logic that needs no device: argument checks of the
class mirror that run before any GPU call, weight handling, sharding, harness."""
import numpy as np
import pytest

from direct_data_driven_mpc_amd import _lib as L
from direct_data_driven_mpc_amd.direct_data_driven_mpc_controller import (
    DataDrivenMPCType, DirectDataDrivenMPCController, SlackVarConstraintTypes)
from direct_data_driven_mpc_amd.distributed import shard_bounds
from direct_data_driven_mpc_amd.engine import _weights
from direct_data_driven_mpc_amd.harness import controller_params, generate_batch


def _kw(**over):
    rng = np.random.default_rng(1)
    kw = dict(n=4, m=2, p=2, u_d=rng.uniform(-1, 1, (400, 2)), y_d=rng.uniform(-1, 1, (400, 2)), L=30,
              Q=3 * np.eye(60), R=1e-4 * np.eye(60), u_s=np.ones((2, 1)), y_s=np.array([[0.65], [0.77]]),
              eps_max=0.002, lamb_alpha=50.0, lamb_sigma=1000.0, c=1.0,
              slack_var_constraint_type=SlackVarConstraintTypes.NONE, controller_type=DataDrivenMPCType.ROBUST)
    kw.update(over)
    return kw


def test_enum_quirks_match_reference():
    # direct_data_driven_mpc_controller.py:11-20 (trailing commas make tuple values)
    assert DataDrivenMPCType.NOMINAL.value == (0,) and DataDrivenMPCType.ROBUST.value == 1
    assert SlackVarConstraintTypes.NON_CONVEX.value == (0,) and SlackVarConstraintTypes.CONVEX.value == (1,)
    assert SlackVarConstraintTypes.NONE.value == 2


def test_constructor_rejections_before_any_device_work():
    with pytest.raises(ValueError, match="Unsupported controller type."):
        DirectDataDrivenMPCController(**_kw(controller_type="robust"))
    with pytest.raises(ValueError, match="Unsupported slack variable constraint type."):
        DirectDataDrivenMPCController(**_kw(slack_var_constraint_type=2))
    with pytest.raises(ValueError, match="All robust MPC parameters"):
        DirectDataDrivenMPCController(**_kw(lamb_sigma=None))
    with pytest.raises(ValueError, match="should match the number of inputs"):
        DirectDataDrivenMPCController(**_kw(m=3))
    with pytest.raises(ValueError, match="required minimum N is 113, but got 100"):
        kw = _kw()
        kw["u_d"], kw["y_d"] = kw["u_d"][:100], kw["y_d"][:100]
        DirectDataDrivenMPCController(**kw)


def test_compat_import_paths():
    from direct_data_driven_mpc.direct_data_driven_mpc_controller import DirectDataDrivenMPCController as C2
    from direct_data_driven_mpc.utilities.hankel_matrix import evaluate_persistent_excitation, hankel_matrix
    assert C2 is DirectDataDrivenMPCController
    assert callable(hankel_matrix) and callable(evaluate_persistent_excitation)


def test_weight_handling():
    assert _weights(3.0, 60, "Q")[0] == L.WEIGHT_SCALAR
    k, v = _weights(3 * np.eye(60), 60, "Q")
    assert k == L.WEIGHT_SCALAR and v.tolist() == [3.0]
    d = np.linspace(1, 2, 60)
    k, v = _weights(np.diag(d), 60, "Q")
    assert k == L.WEIGHT_DIAG and np.array_equal(v, d)
    k, v = _weights(np.eye(60) + 0.1, 60, "Q")                  # non-diagonal: travels as the dense matrix
    assert k == L.WEIGHT_DENSE and v.shape == (60, 60)
    from direct_data_driven_mpc_amd.engine import _expand_weight
    assert np.array_equal(_expand_weight(L.WEIGHT_SCALAR, np.array([2.0]), 3, L.WEIGHT_DENSE), 2.0 * np.eye(3))
    assert np.array_equal(_expand_weight(L.WEIGHT_DIAG, np.array([1.0, 2.0]), 2, L.WEIGHT_DENSE), np.diag([1.0, 2.0]))
    assert np.array_equal(_expand_weight(L.WEIGHT_SCALAR, np.array([2.0]), 3, L.WEIGHT_DIAG), np.full(3, 2.0))
    with pytest.raises(ValueError):
        _weights(np.eye(59), 60, "Q")


def test_shard_bounds_cover_batch_exactly():
    for total in (1, 7, 4096, 262144, 10):
        for world in (1, 2, 3, 8):
            edges = [shard_bounds(total, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == total
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(4, 2, 2)


def test_harness_parameters_and_batched_generation(golden):
    cfg = controller_params()
    assert cfg["lamb_alpha"] == pytest.approx(50.0) and cfg["c"] == 1.0 and cfg["n_mpc_step"] == 4
    assert cfg["robust"] and cfg["slack"] == "none"
    assert controller_params(dict(epsilon_bar=0.0))["lamb_alpha"] == 1000.0
    d = generate_batch(range(5))
    for s in range(5):
        assert np.array_equal(d["u_d"][s], golden[f"s{s}_u_d"])          # RNG stream identical
        assert np.max(np.abs(d["y_d"][s] - golden[f"s{s}_y_d"])) < 1e-14  # batched matmul: last-ulp only
        assert np.max(np.abs(d["x_0"][s] - golden[f"s{s}_x0"])) < 1e-14


def test_yaml_config_helpers(tmp_path):
    # same YAML keys and error behaviour as utilities/yaml_config_loading.py:6-37 and
    # the parameter derivation of utilities/controller/controller_creation.py:105-168
    import os
    from direct_data_driven_mpc_amd import harness as hs
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    model = os.path.join(root, "examples", "config", "models", "four_tank_system_params.yaml")
    ctrl = os.path.join(root, "examples", "config", "controllers", "data_driven_mpc_example_params.yaml")
    pl = hs.plant_from_yaml(model, "FourTankSystem")
    for k in ("A", "B", "C", "D"):
        assert np.array_equal(pl[k], hs.FOUR_TANK[k])
    assert pl["eps_max"] == hs.FOUR_TANK["eps_max"]
    cfg = hs.controller_params_from_yaml(ctrl, "data_driven_mpc_params", m=2, p=2)
    ref = hs.controller_params()
    assert set(cfg) == set(ref)
    for k in ref:
        assert np.all(np.asarray(cfg[k] == ref[k])), k
    assert cfg["lamb_alpha"] == pytest.approx(0.1 / 0.002) and cfg["n_mpc_step"] == cfg["n"] == 4 and cfg["c"] == 1.0
    over = hs.controller_params_from_yaml(ctrl, overrides=dict(controller_type=0, slack_var_constraint_type=1,
                                                               epsilon_bar=0.0))
    assert over["robust"] is False and over["slack"] == "convex" and over["lamb_alpha"] == 1000.0
    with pytest.raises(FileNotFoundError):
        hs.load_yaml_config_params(str(tmp_path / "missing.yaml"), "x")
    with pytest.raises(ValueError, match="Missing `nope` value"):
        hs.load_yaml_config_params(ctrl, "nope")
    with pytest.raises(ValueError):
        hs.controller_params_from_yaml(ctrl, m=3, p=2)


def test_reproduction_start_helpers(golden):
    # equilibrium / observer helpers of the reproduction flow (utilities/initial_state_estimation.py,
    # utilities/reproduction/paper_reproduction.py:80-116) against values captured from the reference
    from direct_data_driven_mpc_amd import harness as hs
    u_eq = hs.equilibrium_input_from_output(hs.FOUR_TANK, [0.4, 0.4])
    assert np.allclose(u_eq, golden["eq_u"], atol=1e-12)
    x_eq = hs.initial_state_from_trajectory(hs.FOUR_TANK, np.tile(u_eq, 4), np.tile([0.4, 0.4], 4))
    assert np.allclose(x_eq, [0.4, 0.4, 0.57975222, 0.4778383], atol=1e-8)
    rngs = [np.random.default_rng(s) for s in (1, 2)]
    x_start, U_n, Y_n = hs.reproduction_start(hs.FOUR_TANK, rngs, [0.4, 0.4], [1.0, 1.0], 4)
    assert x_start.shape == (2, 4) and U_n.shape == (2, 8) and Y_n.shape == (2, 8)
    assert np.allclose(Y_n[:, :2], 0.4, atol=0.0021)                                 # first output = y_0 + noise


def test_example_plot_helper(tmp_path):
    # the minimal plotting of the batched examples (median + band + instance 0); needs matplotlib only
    import os, sys
    pytest.importorskip("matplotlib")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "examples"))
    from _plot import plot_closed_loops
    rng = np.random.default_rng(0)
    u = 1.0 + 0.1 * rng.normal(size=(16, 50, 2)); y = 0.7 + 0.01 * rng.normal(size=(16, 50, 2))
    ydiv = y.copy(); ydiv[:, 30:, :] = np.nan                     # a stopped instance is NaN from there on
    out = tmp_path / "p.png"
    plot_closed_loops(str(out), {"a": (u, y), "b": (u * 2, ydiv)}, [1.0, 1.0], [0.65, 0.77], t0=4, title="t")
    assert out.stat().st_size > 10000
