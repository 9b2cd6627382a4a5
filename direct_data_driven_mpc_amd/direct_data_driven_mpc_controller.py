"""`DirectDataDrivenMPCController` on the MI355X engine.

Same class surface as the reference controller
(direct_data_driven_mpc/direct_data_driven_mpc_controller.py:22-982): same
constructor signature, attributes, methods, exceptions and status strings, so
the reference's harness and example scripts can import it unchanged.  The QP the
reference builds with CVXPY each step (:404-407) is solved by the batched HIP
engine (batch = 1 here; see `engine.BatchedDDMPC` for the batched API).

There is no CPU path: constructing a controller without a HIP device raises.
"""
from __future__ import annotations

from enum import Enum
from typing import List, Optional

import numpy as np

from . import _lib as L
from .engine import BatchedDDMPC
from .utilities.hankel_matrix import evaluate_persistent_excitation, hankel_matrix


class DataDrivenMPCType(Enum):
    # Values mirror the reference enum, including its trailing-comma tuples
    # (controller.py:11-14); callers only use identity / .name.
    NOMINAL = 0,
    ROBUST = 1


class SlackVarConstraintTypes(Enum):
    # controller.py:16-20
    NON_CONVEX = 0,
    CONVEX = 1,
    NONE = 2


class _Value:
    """Stand-in for a cp.Variable: exposes `.value` (numpy column) and `.shape`."""

    def __init__(self, owner, name, rows):
        self._owner, self._name, self.shape = owner, name, (rows, 1)

    @property
    def value(self):
        return self._owner._fetch_solution(self._name)


class _Problem:
    """Stand-in for cp.Problem: `.status`, `.value`, `.solve()`."""

    def __init__(self, owner):
        self._owner = owner
        self.status = None
        self.value = None

    def solve(self, *args, **kwargs):
        self._owner._solve_on_device()
        return self.value


class DirectDataDrivenMPCController:
    def __init__(
        self,
        n: int,
        m: int,
        p: int,
        u_d: np.ndarray,
        y_d: np.ndarray,
        L: int,
        Q: np.ndarray,
        R: np.ndarray,
        u_s: np.ndarray,
        y_s: np.ndarray,
        eps_max: Optional[float] = None,
        lamb_alpha: Optional[float] = None,
        lamb_sigma: Optional[float] = None,
        c: Optional[float] = None,
        slack_var_constraint_type: SlackVarConstraintTypes = SlackVarConstraintTypes.CONVEX,
        controller_type: DataDrivenMPCType = DataDrivenMPCType.NOMINAL,
        n_mpc_step: int = 1,
        use_terminal_constraint: bool = True,
        device: int = 0,
    ):
        # Order of checks follows controller.py:161-240.
        self.controller_type = controller_type
        if controller_type not in (DataDrivenMPCType.NOMINAL, DataDrivenMPCType.ROBUST):
            raise ValueError("Unsupported controller type.")                       # :168
        self.n, self.m, self.p = n, m, p
        self.u_d, self.y_d = u_d, y_d                                              # kept by reference, :177-178
        self.N = u_d.shape[0]
        self.u_past = u_d[-n:, :].reshape(-1, 1)                                   # :184
        self.y_past = y_d[-n:, :].reshape(-1, 1)                                   # :185
        self.L, self.Q, self.R = L, Q, R
        self.u_s, self.y_s = u_s, y_s
        self.eps_max, self.lamb_alpha, self.lamb_sigma, self.c = eps_max, lamb_alpha, lamb_sigma, c
        self.slack_var_constraint_type = slack_var_constraint_type
        if slack_var_constraint_type not in (SlackVarConstraintTypes.NON_CONVEX,
                                             SlackVarConstraintTypes.CONVEX,
                                             SlackVarConstraintTypes.NONE):
            raise ValueError("Unsupported slack variable constraint type.")       # :215
        if self.controller_type == DataDrivenMPCType.ROBUST:
            if None in (eps_max, lamb_alpha, lamb_sigma, c):                        # :219-222
                raise ValueError("All robust MPC parameters (eps_max, lamb_alpha, lamb_sigma, c) must be "
                                 "provided for a 'ROBUST' controller.")
        self.n_mpc_step = n_mpc_step
        self.use_terminal_constraint = use_terminal_constraint
        self._device = device
        self._engine: Optional[BatchedDDMPC] = None
        self._solution_cache = {}
        self.optimal_u = None

        self.evaluate_input_persistent_excitation()
        self.check_prediction_horizon_length()
        self.check_weighting_matrices_dimensions()
        self.initialize_data_driven_mpc()

    # ---- validation (controller.py:242-343) ----------------------------------
    def evaluate_input_persistent_excitation(self) -> None:
        u_d_n = self.u_d.shape[1]
        if u_d_n != self.m:                                                        # :268
            raise ValueError("The length of the elements of the data "
                             f"sequence ({u_d_n}) should match the number of "
                             f"inputs of the system ({self.m}).")
        N_min = self.m * (self.L + 2 * self.n) + self.L + 2 * self.n - 1           # :275
        if self.N < N_min:
            raise ValueError(
                "Initial input trajectory data is not persistently exciting "
                "of order (L + 2 * n). It does not satisfy the inequality: "
                "N - L - 2 * n + 1 ≥ m * (L + 2 * n). The required minimum N "
                f"is {N_min}, but got {self.N}.")
        expected_order = self.L + 2 * self.n
        in_hankel_rank, in_pers_exc = evaluate_persistent_excitation(X=self.u_d, order=expected_order)
        if not in_pers_exc:                                                        # :291-296
            raise ValueError(
                "Initial input trajectory data is not persistently exciting "
                "of order (L + 2 * n). The rank of its induced Hankel matrix "
                f"({in_hankel_rank}) does not match the expected rank ("
                f"{u_d_n * expected_order}).")

    def check_prediction_horizon_length(self) -> None:
        if self.controller_type == DataDrivenMPCType.NOMINAL:
            if self.L < self.n:                                                    # :317-320
                raise ValueError("The prediction horizon (`L`) must be "
                                 "greater than or equal to the estimated "
                                 "system order `n`.")
        elif self.controller_type == DataDrivenMPCType.ROBUST:
            if self.L < 2 * self.n:                                                # :322-325
                raise ValueError("The prediction horizon (`L`) must be "
                                 "greater than or equal to two times the "
                                 "estimated system order `n`.")

    def check_weighting_matrices_dimensions(self) -> None:
        if self.Q.shape != (self.p * self.L, self.p * self.L):                     # :338-340
            raise ValueError("Output weighting square matrix Q should be"
                             "of order (p * L)")
        if self.R.shape != (self.m * self.L, self.m * self.L):                     # :341-343
            raise ValueError("Input weighting square matrix R should be"
                             "of order (m * L)")

    # ---- problem definition ----------------------------------------------------
    def initialize_data_driven_mpc(self) -> None:
        """controller.py:345-387: Hankel matrices, problem definition, first solve."""
        self.HLn_ud = hankel_matrix(self.u_d, self.L + self.n)                     # :376
        self.HLn_yd = hankel_matrix(self.y_d, self.L + self.n)                     # :377
        self.define_optimization_variables()
        self.define_mpc_constraints()
        self.define_cost_function()
        self.define_mpc_problem()
        self.solve_mpc_problem()
        self.get_optimal_control_input()

    def update_and_solve_data_driven_mpc(self) -> None:
        """Per-step entry point, controller.py:389-407."""
        self.define_mpc_constraints()
        self.define_mpc_problem()
        self.solve_mpc_problem()
        self.get_optimal_control_input()

    def define_optimization_variables(self) -> None:
        Ln = self.L + self.n
        self.alpha = _Value(self, "alpha", self.N - Ln + 1)                        # :434
        self.ubar = _Value(self, "ubar", Ln * self.m)                              # :436
        self.ybar = _Value(self, "ybar", Ln * self.p)                              # :438
        if self.controller_type == DataDrivenMPCType.ROBUST:
            self.sigma = _Value(self, "sigma", Ln * self.p)                        # :445

    def define_mpc_constraints(self) -> None:
        """The constraint set is implicit in the engine; this validates the slack type
        exactly where the reference does (controller.py:495-498,664-670)."""
        self.dynamics_constraint = self.define_system_dynamic_constraint()
        self.internal_state_constraint = self.define_internal_state_constraint()
        self.terminal_constraint = (self.define_terminal_state_constraint(u_s=self.u_s, y_s=self.y_s)
                                    if self.use_terminal_constraint else [])
        self.slack_var_constraint = (self.define_slack_variable_constraint()
                                     if self.controller_type == DataDrivenMPCType.ROBUST else [])
        self.constraints = (self.dynamics_constraint + self.internal_state_constraint +
                            self.terminal_constraint + self.slack_var_constraint)

    def define_system_dynamic_constraint(self) -> List:
        return ["dynamics: [ubar; ybar(+sigma)] == [HLn_ud; HLn_yd] @ alpha"]      # :536-545

    def define_internal_state_constraint(self) -> List:
        return ["internal state: [ubar[:n*m]; ybar[:n*p]] == [u_past; y_past]"]    # :577-581

    def define_terminal_state_constraint(self, u_s: np.ndarray, y_s: np.ndarray) -> List:
        return ["terminal: [ubar[L*m:]; ybar[L*p:]] == [tile(u_s, n); tile(y_s, n)]"]   # :612-627

    def define_slack_variable_constraint(self) -> List:
        if self.slack_var_constraint_type == SlackVarConstraintTypes.NON_CONVEX:   # :664-670
            raise NotImplementedError(
                "Robust Data-Driven MPC with a Non-Convex slack variable "
                "constraint is not currently implemented, since it cannot "
                "be efficiently solved.")
        if self.slack_var_constraint_type == SlackVarConstraintTypes.CONVEX:       # :671-675
            return ["slack: norm(sigma[n*p:], inf) <= c * eps_max"]
        return []

    def define_cost_function(self) -> None:
        self.cost = "quad_form(ubar_pred - u_s, R) + quad_form(ybar_pred - y_s, Q)"   # :708-710
        if self.controller_type == DataDrivenMPCType.ROBUST:
            self.cost += " + lamb_alpha*eps_max*|alpha|^2 + lamb_sigma*|sigma|^2"     # :714-716

    def define_mpc_problem(self) -> None:
        """(Re)creates the engine handle when the parameters changed (controller.py:724-737)."""
        if self._engine is None:
            robust = self.controller_type == DataDrivenMPCType.ROBUST
            slack = {SlackVarConstraintTypes.NON_CONVEX: L.SLACK_NON_CONVEX,
                     SlackVarConstraintTypes.CONVEX: L.SLACK_CONVEX,
                     SlackVarConstraintTypes.NONE: L.SLACK_NONE}[self.slack_var_constraint_type]
            self._engine = BatchedDDMPC(
                n=self.n, m=self.m, p=self.p, L_=self.L, N=self.N, Q=self.Q, R=self.R,
                u_s=self.u_s, y_s=self.y_s, batch=1,
                controller_type=L.ROBUST if robust else L.NOMINAL, slack_type=slack,
                eps_max=self.eps_max, lamb_alpha=self.lamb_alpha, lamb_sigma=self.lamb_sigma, c=self.c,
                use_terminal_constraint=self.use_terminal_constraint, device=self._device)
            self._engine.set_data(np.asarray(self.u_d, dtype=np.float64)[None],
                                  np.asarray(self.y_d, dtype=np.float64)[None])
        if not hasattr(self, "problem") or self.problem is None:
            self.problem = _Problem(self)

    # ---- solve -----------------------------------------------------------------
    def _solve_on_device(self) -> None:
        up = np.asarray(self.u_past, dtype=np.float64).reshape(1, -1)
        yp = np.asarray(self.y_past, dtype=np.float64).reshape(1, -1)
        # The construction-time solve is a cold one; later control steps only change u_past / y_past
        # (controller.py:404-407), so they go through ddmpc_step: the affine law prepared once per data
        # set when the QP has no inequality, a cold solve otherwise.  `use_warm_steps = False` forces
        # cold solves everywhere.
        warm = bool(getattr(self, "use_warm_steps", True)) and getattr(self, "_cold_solved", False)
        u_opt, cost, status, _ = self._engine.solve(up, yp, warm=warm)
        self._cold_solved = True
        self._solution_cache = {}
        self._last_u = u_opt[0].copy()
        self.problem.status = L.STATUS_STRINGS.get(int(status[0]), "solver_error")
        self.problem.value = float(cost[0])

    def _fetch_solution(self, name):
        if self.problem is None or self.problem.status is None:
            return None
        if name not in self._solution_cache:
            self._solution_cache[name] = self._engine.get_solution(name)[0].reshape(-1, 1)
        return self._solution_cache[name]

    def solve_mpc_problem(self) -> str:
        self.problem.solve()                                                       # :753
        return self.problem.status

    def get_problem_solve_status(self) -> str:
        return self.problem.status                                                 # :767

    def get_optimal_cost_value(self) -> float:
        return self.problem.value                                                  # :778

    def get_optimal_control_input(self) -> np.ndarray:
        if self.problem.status in ["optimal", "optimal_inaccurate"]:              # :804
            self.optimal_u = self._last_u.copy()
            return self.optimal_u
        raise ValueError("MPC problem was not solved optimally.")                  # :808

    def get_optimal_control_input_at_step(self, n_step: int = 0) -> np.ndarray:
        if not 0 <= n_step < self.L:                                               # :834-837
            raise ValueError(
                f"The specified prediction time step ({n_step}) is out of "
                f"range. It should be within [0, {self.L - 1}].")
        return self.optimal_u[n_step * self.m:(n_step + 1) * self.m]               # :839

    # ---- past-window bookkeeping (controller.py:844-943) ---------------------
    def store_input_output_measurement(self, u_current: np.ndarray, y_current: np.ndarray) -> None:
        expected_u0_dim = (self.m, 1)
        expected_y0_dim = (self.p, 1)
        if u_current.shape != expected_u0_dim or y_current.shape != expected_y0_dim:    # :882-888
            raise ValueError(
                f"Incorrect dimensions. Expected dimensions are "
                f"{expected_u0_dim} for u_current and {expected_y0_dim} for "
                f"y_current, but got {u_current.shape} and "
                f"{y_current.shape} instead.")
        self.u_past = np.vstack([self.u_past[self.m:], u_current])                 # :893
        self.y_past = np.vstack([self.y_past[self.p:], y_current])                 # :895

    def set_past_input_output_data(self, u_past: np.ndarray, y_past: np.ndarray) -> None:
        expected_u_dim = (self.n * self.m, 1)
        expected_y_dim = (self.n * self.p, 1)
        if u_past.shape != expected_u_dim:                                         # :930-933
            raise ValueError(
                f"Incorrect dimensions. u_past must be shaped as "
                f"{expected_u_dim}. Got {u_past.shape}. instead")
        if y_past.shape != expected_y_dim:                                         # :934-937
            raise ValueError(
                f"Incorrect dimensions. y_past must be shaped as "
                f"{expected_y_dim}. Got {y_past.shape} instead.")
        self.u_past = u_past
        self.y_past = y_past

    def set_input_output_setpoints(self, u_s: np.ndarray, y_s: np.ndarray) -> None:
        if u_s.shape != self.u_s.shape:                                            # :970-972
            raise ValueError(f"Incorrect dimensions. u_s must have shape "
                             f"{self.u_s.shape}, got {u_s.shape}")
        if y_s.shape != self.y_s.shape:                                            # :973-975
            raise ValueError(f"Incorrect dimensions. y_s must have shape "
                             f"{self.y_s.shape}, got {y_s.shape}")
        self.u_s = u_s
        self.y_s = y_s
        if self._engine is not None:
            self._engine.set_setpoints(u_s, y_s)
        self.initialize_data_driven_mpc()                                          # :982
