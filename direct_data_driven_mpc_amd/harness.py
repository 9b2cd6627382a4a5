"""Host-side harness: benchmark parameters and batched synthetic-input generation.

Not part of the QP hot path.  It reproduces, vectorised over a batch of seeds,
the reference's RNG draw order for one controller instance
(utilities/controller/controller_operation.py:59-75,126-133 and
examples/direct_data_driven_mpc_example.py:282-287) so that instance `i` of a
batch is the same problem the reference example would build with `--seed i`.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import numpy as np

# examples/config/models/four_tank_system_params.yaml:10-26
FOUR_TANK = dict(
    A=np.array([[0.921, 0, 0.041, 0], [0, 0.918, 0, 0.033], [0, 0, 0.924, 0], [0, 0, 0, 0.937]], float),
    B=np.array([[0.017, 0.001], [0.001, 0.023], [0, 0.061], [0.072, 0]], float),
    C=np.array([[1, 0, 0, 0], [0, 1, 0, 0]], float),
    D=np.zeros((2, 2)),
    eps_max=0.002,
)

# examples/config/controllers/data_driven_mpc_example_params.yaml:10-22
EXAMPLE_PARAMS = dict(
    N=400, u_d_range=(-1.0, 1.0), epsilon_bar=0.002, L=30, Q_scalar=3.0, R_scalar=1e-4,
    lambda_sigma=1000.0, lambda_alpha_epsilon_bar=0.1, slack_var_constraint_type=0,
    controller_type=1, n=4, u_s=(1.0, 1.0), y_s=(0.65, 0.77),
)


def controller_params(overrides: Optional[Dict] = None) -> Dict:
    """Parameter derivation of utilities/controller/controller_creation.py:105-168:
    Q = q*I, R = r*I, lamb_alpha = lambda_alpha_eps / eps (1000 if eps == 0), c = 1,
    n_mpc_step = n; slack map {0: NONE, 1: CONVEX, 2: NON_CONVEX}, type map {0: NOMINAL, 1: ROBUST}."""
    d = dict(EXAMPLE_PARAMS)
    if overrides:
        d.update(overrides)
    eps = d["epsilon_bar"]
    m, p = len(d["u_s"]), len(d["y_s"])
    return dict(
        n=d["n"], m=m, p=p, L=d["L"], N=d["N"], u_range=tuple(d["u_d_range"]),
        Q=d["Q_scalar"], R=d["R_scalar"], eps_max=eps,
        lamb_alpha=(d["lambda_alpha_epsilon_bar"] / eps if eps != 0 else 1000.0),
        lamb_sigma=d["lambda_sigma"], c=1.0,
        u_s=np.array(d["u_s"], float), y_s=np.array(d["y_s"], float),
        robust=bool(d["controller_type"]), slack={0: "none", 1: "convex", 2: "non_convex"}[d["slack_var_constraint_type"]],
        n_mpc_step=d["n"], tec=d.get("tec", True),
    )


def load_yaml_config_params(config_file: str, key: str):
    """Same contract as utilities/yaml_config_loading.py:6-37: FileNotFoundError for a missing file,
    ValueError for a missing key."""
    import os
    import yaml
    if not os.path.exists(config_file):
        raise FileNotFoundError(f"Configuration file {config_file} not found.")
    with open(config_file, "r") as fh:
        config = yaml.safe_load(fh)
    if key not in config:
        raise ValueError(f"Missing `{key}` value in the configuration file.")
    return config[key]


def plant_from_yaml(config_file: str, key: str = "FourTankSystem") -> Dict:
    """Model file -> dict(A, B, C, D, eps_max) (utilities/model_simulation.py:161-195 reads the same keys)."""
    d = load_yaml_config_params(config_file, key)
    return dict(A=np.array(d["A"], float), B=np.array(d["B"], float), C=np.array(d["C"], float),
                D=np.array(d["D"], float), eps_max=float(d["eps_max"]))


def controller_params_from_yaml(config_file: str, key: str = "data_driven_mpc_params", m: Optional[int] = None,
                                p: Optional[int] = None, overrides: Optional[Dict] = None) -> Dict:
    """Controller file -> derived parameters (controller_creation.py:105-168).  `m`, `p` (from the model)
    are checked against the setpoint lengths like the reference does when it reshapes u_s / y_s."""
    d = dict(load_yaml_config_params(config_file, key))
    if overrides:
        d.update(overrides)
    if m is not None and len(d["u_s"]) != m:
        raise ValueError("u_s must have m entries")
    if p is not None and len(d["y_s"]) != p:
        raise ValueError("y_s must have p entries")
    return controller_params(d)


def _observer(A, B, C, D):
    """pinv(O) and T of the least-squares initial-state observer
    (utilities/initial_state_estimation.py:3-24,72-93,131)."""
    ns, m, p = A.shape[0], B.shape[1], C.shape[0]
    O = np.vstack([C @ np.linalg.matrix_power(A, i) for i in range(ns)])
    T = np.zeros((p * ns, m * ns))
    for i in range(ns):
        for j in range(i + 1):
            blk = D if i == j else C @ np.linalg.matrix_power(A, i - j - 1) @ B
            T[i * p:(i + 1) * p, j * m:(j + 1) * m] = blk
    return np.linalg.pinv(O), T


def simulate_batch(A, B, C, D, x, U, W):
    """x+ = Ax + Bu, y = Cx + Du + w with the output taken before the state update
    (utilities/model_simulation.py:93-98), for a batch: x [Bt,ns], U [Bt,T,m], W [Bt,T,p]."""
    Bt, T = U.shape[0], U.shape[1]
    Y = np.empty((Bt, T, C.shape[0]))
    for k in range(T):
        Y[:, k] = x @ C.T + U[:, k] @ D.T + W[:, k]
        x = x @ A.T + U[:, k] @ B.T
    return Y, x


def generate_batch(seeds: Sequence[int], N: int = 400, plant: Optional[Dict] = None,
                   u_range=(-1.0, 1.0)) -> Dict[str, np.ndarray]:
    """Per-seed draws in the reference order, plant simulation batched.
    Returns u_d [B,N,m], y_d [B,N,p], x_0 [B,ns], x_end [B,ns] (state after the data run)."""
    pl = plant or FOUR_TANK
    A, Bm, C, D, eps = pl["A"], pl["B"], pl["C"], pl["D"], pl["eps_max"]
    ns, m, p = A.shape[0], Bm.shape[1], C.shape[0]
    nb = len(seeds)
    x_i0 = np.empty((nb, ns)); u_i = np.empty((nb, ns, m)); w_i = np.empty((nb, ns, p))
    u_d = np.empty((nb, N, m)); w_d = np.empty((nb, N, p))
    rngs = []
    for b, s in enumerate(seeds):
        rng = np.random.default_rng(int(s))
        x_i0[b] = rng.uniform(-1.0, 1.0, size=ns)
        u_i[b] = rng.uniform(*u_range, (ns, m))
        w_i[b] = eps * rng.uniform(-1.0, 1.0, (ns, p))
        u_d[b] = rng.uniform(*u_range, (N, m))
        w_d[b] = eps * rng.uniform(-1.0, 1.0, (N, p))
        rngs.append(rng)
    y_i, _ = simulate_batch(A, Bm, C, D, x_i0, u_i, w_i)
    Opinv, T = _observer(A, Bm, C, D)
    x_0 = (y_i.reshape(nb, -1) - u_i.reshape(nb, -1) @ T.T) @ Opinv.T
    y_d, x_end = simulate_batch(A, Bm, C, D, x_0, u_d, w_d)
    return dict(u_d=u_d, y_d=y_d, x_0=x_0, x_end=x_end, rngs=rngs)


def equilibrium_input_from_output(plant: Dict, y_eq) -> np.ndarray:
    """Input holding the output at `y_eq` in steady state: u = pinv(C (I - A)^-1 B + D) y
    (utilities/initial_state_estimation.py:171-204)."""
    A, Bm, C, D = plant["A"], plant["B"], plant["C"], plant["D"]
    M = C @ np.linalg.inv(np.eye(A.shape[0]) - A) @ Bm + D
    return np.linalg.pinv(M) @ np.asarray(y_eq, dtype=float)


def initial_state_from_trajectory(plant: Dict, U, Y) -> np.ndarray:
    """Least-squares observer x0 = pinv(O)(Y - T U) from `ns` steps of inputs U [ns*m] and outputs Y [ns*p]
    (utilities/initial_state_estimation.py:3-24,72-93,131)."""
    Opinv, T = _observer(plant["A"], plant["B"], plant["C"], plant["D"])
    return Opinv @ (np.asarray(Y, dtype=float).reshape(-1) - T @ np.asarray(U, dtype=float).reshape(-1))


def reproduction_start(plant: Dict, rngs, y_0, u_s, n: int):
    """Start of the paper-reproduction runs for a batch of instances (utilities/reproduction/
    paper_reproduction.py:80-116 and utilities/controller/controller_operation.py:190-197): the plant is put at
    the equilibrium of output `y_0`, then the input setpoint `u_s` is applied for `n` noisy steps (noise drawn
    from each instance's generator, after the data generation).  Returns (x_start [B,ns], U_n [B,n*m], Y_n [B,n*p])."""
    Bt = len(rngs)
    m, p = plant["B"].shape[1], plant["C"].shape[0]
    u_eq = equilibrium_input_from_output(plant, y_0)
    x_eq = initial_state_from_trajectory(plant, np.tile(u_eq, plant["A"].shape[0]), np.tile(np.asarray(y_0, float), plant["A"].shape[0]))
    U = np.tile(np.asarray(u_s, dtype=float), (Bt, n, 1))
    W = np.stack([plant["eps_max"] * rng.uniform(-1.0, 1.0, (n, p)) for rng in rngs])
    Y, x_start = simulate_batch(plant["A"], plant["B"], plant["C"], plant["D"], np.tile(x_eq, (Bt, 1)), U, W)
    return x_start, U.reshape(Bt, n * m), Y.reshape(Bt, n * p)


def step_report_line(t: int, cost: float, u_s, y_s, u_k, y_k) -> str:
    """The per-solve report line of the reference's control loop at verbosity 2
    (utilities/controller/controller_operation.py:310-329): solve time step, cost, setpoint errors of the
    last applied step."""
    u_error = np.asarray(u_s, float).flatten() - np.asarray(u_k, float).flatten()
    y_error = np.asarray(y_s, float).flatten() - np.asarray(y_k, float).flatten()
    ue = ', '.join([f'u_{i + 1}e = {e:>6.3f}' for i, e in enumerate(u_error)])
    ye = ', '.join([f'y_{i + 1}e = {e:>6.3f}' for i, e in enumerate(y_error)])
    return f"    Time step: {t:>4} - MPC cost value: {cost:>8.4f} - Error: {ue}, {ye}"


def simulate_control_loop(plant: Dict, x0, controller, n_steps: int, rng, verbose: int = 0):
    """Drive ONE controller object in closed loop with the plant, the way the reference's
    `simulate_data_driven_mpc_control_loop` does (utilities/controller/controller_operation.py:201-331):
    noise drawn first (`eps_max * uniform(-1, 1, (n_steps, p))`), then per solve: update_and_solve ->
    for each of the next n_mpc_step steps: apply optimal_u[k - t], y = Cx + Du + w before x <- Ax + Bu
    (utilities/model_simulation.py:93-98), store the measurement.  `controller` is any object with the
    reference class's per-step methods.  Returns (u_sys, y_sys, x_end)."""
    A, Bm, C, D, eps = plant["A"], plant["B"], plant["C"], plant["D"], plant["eps_max"]
    m, p = Bm.shape[1], C.shape[0]
    x = np.array(x0, dtype=float).reshape(-1)
    n_mpc_step = controller.n_mpc_step
    u_sys = np.zeros((n_steps, m)); y_sys = np.zeros((n_steps, p))
    w_sys = eps * rng.uniform(-1.0, 1.0, (n_steps, p))
    for t in range(0, n_steps, n_mpc_step):
        controller.update_and_solve_data_driven_mpc()
        for k in range(t, min(t + n_mpc_step, n_steps)):
            u_sys[k, :] = controller.get_optimal_control_input_at_step(n_step=k - t)
            y_sys[k, :] = C @ x + D @ u_sys[k, :] + w_sys[k, :]
            x = A @ x + Bm @ u_sys[k, :]
            controller.store_input_output_measurement(u_current=u_sys[k, :].reshape(-1, 1),
                                                      y_current=y_sys[k, :].reshape(-1, 1))
        if verbose > 1:
            print(step_report_line(t, controller.get_optimal_cost_value(), controller.u_s, controller.y_s,
                                   u_sys[k, :], y_sys[k, :]))
    return u_sys, y_sys, x
