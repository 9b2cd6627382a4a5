"""CPU oracle: full-space restatement of the reference's Data-Driven MPC QP.

TEST INFRASTRUCTURE ONLY -- never imported by the product package.

Parity status (see DESIGN.md "Oracle"):
  * `hankel_matrix` is pinned by the reference's docstring known-answer
    (direct_data_driven_mpc/utilities/hankel_matrix.py:26-37) and by fixtures
    generated from the reference's own importable modules
    (tests/golden/make_golden.py).
  * The QP solve itself is **parity unpinned** against the reference: the
    arithmetic lives in `cvxpy` (unpinned, setup.py:21), which is not installed
    anywhere in this pipeline (ordinary ModuleNotFoundError) and the reference
    holds no tests/golden vectors for it.  The oracle is therefore anchored on
    (i) a literal full-space restatement of the formulation the reference hands
    to CVXPY (variables alpha/ubar/ybar/sigma exactly as
    direct_data_driven_mpc_controller.py:409-445 lays them out), solved through
    its KKT system, (ii) a solver-independent KKT certificate
    (`kkt_certificate`), and (iii) a third-party cross-check with
    scipy.optimize recorded in tests/golden (make_golden.py).

Everything here works on the *un-reduced* problem (571 variables, 168 equality
rows, 120 box rows for the four-tank L=30/N=400 case) on purpose: the product
solves a reduced r x r system, so a shared algebra mistake cannot hide.
All `file:line` citations are relative to /root/reference.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

import numpy as np

# Status strings are CVXPY's (direct_data_driven_mpc_controller.py:755,804).
OPTIMAL = "optimal"
OPTIMAL_INACCURATE = "optimal_inaccurate"
INFEASIBLE = "infeasible"
SOLVER_ERROR = "solver_error"


# --------------------------------------------------------------------------
# a1 / a2: Hankel helper (direct_data_driven_mpc/utilities/hankel_matrix.py)
# --------------------------------------------------------------------------
def hankel_matrix(X: np.ndarray, L: int) -> np.ndarray:
    """Block-Hankel matrix H[k*nch + ch, i] = X[i + k, ch].

    Follows hankel_matrix.py:5-53 (column i is X[i:i+L].flatten()); raises the
    same ValueError when N < L (hankel_matrix.py:43-44).
    """
    X = np.asarray(X, dtype=float)
    N, nch = X.shape
    if N < L:
        raise ValueError("N must be greater than or equal to L.")
    cols = N - L + 1
    t = np.arange(L)[:, None] + np.arange(cols)[None, :]      # (L, cols) time idx
    return X[t].transpose(0, 2, 1).reshape(L * nch, cols).copy()


def evaluate_persistent_excitation(X: np.ndarray, order: int) -> Tuple[int, bool]:
    """rank(H_order(X)) == nch*order, hankel_matrix.py:55-87 (SVD rank)."""
    nch = X.shape[1]
    rank = int(np.linalg.matrix_rank(hankel_matrix(X, order)))
    return rank, rank == nch * order


# --------------------------------------------------------------------------
# Problem description
# --------------------------------------------------------------------------
@dataclass
class QPSpec:
    """Controller parameters, as the reference constructor receives them
    (direct_data_driven_mpc_controller.py:95-116)."""
    n: int
    m: int
    p: int
    L: int
    Q: np.ndarray                     # (p*L, p*L)
    R: np.ndarray                     # (m*L, m*L)
    u_s: np.ndarray                   # (m,)
    y_s: np.ndarray                   # (p,)
    robust: bool = False              # DataDrivenMPCType.ROBUST
    eps_max: Optional[float] = None
    lamb_alpha: Optional[float] = None
    lamb_sigma: Optional[float] = None
    c: Optional[float] = None
    slack: str = "convex"             # "none" | "convex"  (:631-677)
    tec: bool = True                  # use_terminal_constraint (:489-492)

    @property
    def Ln(self) -> int:
        return self.L + self.n


@dataclass
class FullQP:
    """min x'Px + q'x + const  s.t.  A x = b,  |x[box_idx]| <= bound."""
    P: np.ndarray
    q: np.ndarray
    const: float
    A: np.ndarray
    b: np.ndarray
    box_idx: np.ndarray
    bound: float
    sl: Dict[str, slice] = field(default_factory=dict)


def build_fullspace_qp(spec: QPSpec, u_d: np.ndarray, y_d: np.ndarray,
                       u_past: np.ndarray, y_past: np.ndarray) -> FullQP:
    """Assemble the QP exactly as the reference states it to CVXPY.

    Variable stacking x = [alpha; ubar; ybar; sigma]
    (direct_data_driven_mpc_controller.py:434-445); rows of A in the order
    dynamics (:536-545), internal state (:577-581), terminal (:612-627); cost
    from :703-716; slack box from :659,674.
    """
    n, m, p, L, Ln = spec.n, spec.m, spec.p, spec.L, spec.Ln
    Hu = hankel_matrix(u_d, Ln)                                   # :376
    Hy = hankel_matrix(y_d, Ln)                                   # :377
    c = Hu.shape[1]
    nu, ny = Ln * m, Ln * p
    ns = ny if spec.robust else 0
    nx = c + nu + ny + ns
    s_a = slice(0, c)
    s_u = slice(c, c + nu)
    s_y = slice(c + nu, c + nu + ny)
    s_s = slice(c + nu + ny, nx)

    u_s = np.asarray(spec.u_s, float).reshape(-1)
    y_s = np.asarray(spec.y_s, float).reshape(-1)
    u_past = np.asarray(u_past, float).reshape(-1)
    y_past = np.asarray(y_past, float).reshape(-1)

    # ---- cost: x'Px + q'x + const ------------------------------------
    P = np.zeros((nx, nx))
    q = np.zeros(nx)
    iu = np.arange(c + n * m, c + nu)            # ubar[n*m:]  (:703)
    iy = np.arange(c + nu + n * p, c + nu + ny)  # ybar[n*p:]  (:705)
    us_t = np.tile(u_s, L)
    ys_t = np.tile(y_s, L)
    Rs = 0.5 * (spec.R + spec.R.T)
    Qs = 0.5 * (spec.Q + spec.Q.T)
    P[np.ix_(iu, iu)] += Rs                      # quad_form(ubar_pred-us, R) :709
    q[iu] += -2.0 * Rs @ us_t
    P[np.ix_(iy, iy)] += Qs                      # quad_form(ybar_pred-ys, Q) :710
    q[iy] += -2.0 * Qs @ ys_t
    const = float(us_t @ Rs @ us_t + ys_t @ Qs @ ys_t)
    if spec.robust:                              # :714-716
        ia = np.arange(c)
        P[ia, ia] += spec.lamb_alpha * spec.eps_max
        isg = np.arange(c + nu + ny, nx)
        P[isg, isg] += spec.lamb_sigma

    # ---- equalities ----------------------------------------------------
    rows = []
    rhs = []
    # dynamics: [ubar; ybar (+sigma)] - [Hu;Hy] alpha = 0      (:536-545)
    D = np.zeros((nu + ny, nx))
    D[:nu, s_a] = -Hu
    D[nu:, s_a] = -Hy
    D[:nu, s_u] = np.eye(nu)
    D[nu:, s_y] = np.eye(ny)
    if spec.robust:
        D[nu:, s_s] = np.eye(ny)
    rows.append(D)
    rhs.append(np.zeros(nu + ny))
    # internal state: ubar[:n*m]=u_past, ybar[:n*p]=y_past     (:577-581)
    E = np.zeros((n * m + n * p, nx))
    E[np.arange(n * m), c + np.arange(n * m)] = 1.0
    E[n * m + np.arange(n * p), c + nu + np.arange(n * p)] = 1.0
    rows.append(E)
    rhs.append(np.concatenate([u_past, y_past]))
    if spec.tec:                                               # :612-627
        T = np.zeros((n * m + n * p, nx))
        T[np.arange(n * m), c + L * m + np.arange(n * m)] = 1.0
        T[n * m + np.arange(n * p), c + nu + L * p + np.arange(n * p)] = 1.0
        rows.append(T)
        rhs.append(np.concatenate([np.tile(u_s, n), np.tile(y_s, n)]))
    A = np.vstack(rows)
    b = np.concatenate(rhs)

    # ---- slack box: ||sigma[n*p:]||_inf <= c*eps_max          (:659,674)
    if spec.robust and spec.slack == "convex":
        box_idx = np.arange(c + nu + ny + n * p, nx)
        bound = float(spec.c * spec.eps_max)
    else:
        box_idx = np.zeros(0, dtype=int)
        bound = np.inf
    return FullQP(P=P, q=q, const=const, A=A, b=b, box_idx=box_idx, bound=bound,
                  sl=dict(alpha=s_a, ubar=s_u, ybar=s_y, sigma=s_s))


# --------------------------------------------------------------------------
# Solvers on the full-space problem
# --------------------------------------------------------------------------
def _kkt_solve(P, q, A, b):
    """Stationary point of x'Px+q'x s.t. Ax=b through the dense KKT system."""
    nx, ne = P.shape[0], A.shape[0]
    K = np.zeros((nx + ne, nx + ne))
    K[:nx, :nx] = 2.0 * P
    K[:nx, nx:] = A.T
    K[nx:, :nx] = A
    rhs = np.concatenate([-q, b])
    try:
        sol = np.linalg.solve(K, rhs)
        if not np.all(np.isfinite(sol)):
            raise np.linalg.LinAlgError
    except np.linalg.LinAlgError:
        sol = np.linalg.lstsq(K, rhs, rcond=None)[0]
    return sol[:nx], sol[nx:]


def _kkt_solve_minnorm(P, q, A, b):
    """Same, for the nominal scheme whose KKT matrix is singular in alpha
    (no regulariser, :720-722): minimum-norm least-squares solution."""
    nx, ne = P.shape[0], A.shape[0]
    K = np.zeros((nx + ne, nx + ne))
    K[:nx, :nx] = 2.0 * P
    K[:nx, nx:] = A.T
    K[nx:, :nx] = A
    rhs = np.concatenate([-q, b])
    sol = np.linalg.lstsq(K, rhs, rcond=1e-13)[0]
    return sol[:nx], sol[nx:]


@dataclass
class OracleSolution:
    status: str
    x: np.ndarray
    cost: float
    alpha: np.ndarray
    ubar: np.ndarray
    ybar: np.ndarray
    sigma: Optional[np.ndarray]
    optimal_u: np.ndarray           # ubar[n*m:], (:799-805)
    iters: int
    active: np.ndarray              # signed active set over box_idx


def solve_fullspace(spec: QPSpec, u_d, y_d, u_past, y_past,
                    max_iter: int = 100) -> OracleSolution:
    """Solve the reference QP (`self.problem.solve()`, :753) exactly.

    Equality-only cases: one KKT solve.  Slack CONVEX: primal-dual active-set
    on the full-space KKT system (finite termination at the exact optimum of
    the strictly convex robust problem).
    """
    qp = build_fullspace_qp(spec, u_d, y_d, u_past, y_past)
    nx = qp.P.shape[0]
    nb = qp.box_idx.size
    act = np.zeros(nb, dtype=int)
    status = OPTIMAL
    iters = 0
    if not spec.robust:
        x, _ = _kkt_solve_minnorm(qp.P, qp.q, qp.A, qp.b)
    elif nb == 0:
        x, _ = _kkt_solve(qp.P, qp.q, qp.A, qp.b)
    else:
        x = None
        for iters in range(1, max_iter + 1):
            idx = np.nonzero(act)[0]
            Eb = np.zeros((idx.size, nx))
            Eb[np.arange(idx.size), qp.box_idx[idx]] = 1.0
            A2 = np.vstack([qp.A, Eb])
            b2 = np.concatenate([qp.b, act[idx] * qp.bound])
            x, nu = _kkt_solve(qp.P, qp.q, A2, b2)
            mu = np.zeros(nb)
            mu[idx] = nu[qp.A.shape[0]:]
            sig = x[qp.box_idx]
            new = np.zeros(nb, dtype=int)
            # primal-dual active-set update: keep a bound while its multiplier
            # has the right sign, add a bound when the free value violates it.
            new[(act == 1) & (mu > 0)] = 1
            new[(act == -1) & (mu < 0)] = -1
            new[(act == 0) & (sig > qp.bound)] = 1
            new[(act == 0) & (sig < -qp.bound)] = -1
            if np.array_equal(new, act):
                break
            act = new
        else:
            status = SOLVER_ERROR
    if not np.all(np.isfinite(x)):
        status = SOLVER_ERROR
    cost = float(x @ qp.P @ x + qp.q @ x + qp.const)
    sl = qp.sl
    n, m = spec.n, spec.m
    ubar = x[sl["ubar"]]
    return OracleSolution(
        status=status, x=x, cost=cost, alpha=x[sl["alpha"]], ubar=ubar,
        ybar=x[sl["ybar"]], sigma=(x[sl["sigma"]] if spec.robust else None),
        optimal_u=ubar[n * m:].copy(), iters=iters, active=act)


def kkt_certificate(spec: QPSpec, u_d, y_d, u_past, y_past, x: np.ndarray,
                    act_tol: float = 1e-9) -> Dict[str, float]:
    """Solver-independent optimality certificate for a candidate `x`.

    Returns primal residuals and the best achievable stationarity residual
    (multipliers fitted by least squares on the constraints active at `x`),
    plus the worst wrong-signed bound multiplier.  A true optimum has all of
    them ~ machine precision times the problem's scale.
    """
    qp = build_fullspace_qp(spec, u_d, y_d, u_past, y_past)
    res_eq = float(np.max(np.abs(qp.A @ x - qp.b)))
    sig = x[qp.box_idx] if qp.box_idx.size else np.zeros(0)
    res_box = float(np.max(np.maximum(np.abs(sig) - qp.bound, 0.0))) if sig.size else 0.0
    up = np.nonzero(sig >= qp.bound - act_tol)[0]
    lo = np.nonzero(sig <= -qp.bound + act_tol)[0]
    idx = np.concatenate([up, lo])
    Eb = np.zeros((idx.size, x.size))
    Eb[np.arange(idx.size), qp.box_idx[idx]] = 1.0
    G = np.vstack([qp.A, Eb]).T                       # columns = constraint normals
    g = -(2.0 * qp.P @ x + qp.q)
    mult = np.linalg.lstsq(G, g, rcond=None)[0]
    res_stat = float(np.max(np.abs(G @ mult - g)))
    mu = mult[qp.A.shape[0]:]
    bad = 0.0
    if up.size:
        bad = max(bad, float(np.max(np.maximum(-mu[:up.size], 0.0))))
    if lo.size:
        bad = max(bad, float(np.max(np.maximum(mu[up.size:], 0.0))))
    return dict(res_eq=res_eq, res_box=res_box, res_stat=res_stat, dual_sign=bad,
                grad_scale=float(np.max(np.abs(g))) if g.size else 0.0)


# --------------------------------------------------------------------------
# Plant side (used only to build test inputs and closed loops)
# --------------------------------------------------------------------------
FOUR_TANK = dict(                     # examples/config/models/four_tank_system_params.yaml:10-26
    A=np.array([[0.921, 0, 0.041, 0], [0, 0.918, 0, 0.033],
                [0, 0, 0.924, 0], [0, 0, 0, 0.937]], float),
    B=np.array([[0.017, 0.001], [0.001, 0.023], [0, 0.061], [0.072, 0]], float),
    C=np.array([[1, 0, 0, 0], [0, 1, 0, 0]], float),
    D=np.zeros((2, 2)),
    eps_max=0.002,
)

EXAMPLE_PARAMS = dict(                # examples/config/controllers/data_driven_mpc_example_params.yaml:10-22
    N=400, u_d_range=(-1.0, 1.0), epsilon_bar=0.002, L=30, Q_scalar=3.0,
    R_scalar=1e-4, lambda_sigma=1000.0, lambda_alpha_epsilon_bar=0.1,
    slack_var_constraint_type=0, controller_type=1, n=4,
    u_s=(1.0, 1.0), y_s=(0.65, 0.77),
)


class Plant:
    """x+ = Ax + Bu, y = Cx + Du + w, output *before* the state update
    (utilities/model_simulation.py:93-98)."""

    def __init__(self, A, B, C, D, eps_max=0.0):
        self.A, self.B, self.C, self.D = (np.asarray(M, float) for M in (A, B, C, D))
        self.eps_max = float(eps_max)
        self.ns, self.m, self.p = self.A.shape[0], self.B.shape[1], self.C.shape[0]
        self.x = np.zeros(self.ns)

    def step(self, u, w):
        y = self.C @ self.x + self.D @ u + w
        self.x = self.A @ self.x + self.B @ u
        return y

    def simulate(self, U, W):
        return np.array([self.step(U[k], W[k]) for k in range(U.shape[0])])

    def initial_state_from_trajectory(self, U, Y):
        """Least-squares observer x0 = pinv(O)(Y - T U),
        utilities/initial_state_estimation.py:3-24,72-93,131."""
        t = self.ns
        O = np.vstack([self.C @ np.linalg.matrix_power(self.A, i) for i in range(t)])
        T = np.zeros((self.p * t, self.m * t))
        for i in range(t):
            for j in range(i + 1):
                blk = self.D if i == j else self.C @ np.linalg.matrix_power(self.A, i - j - 1) @ self.B
                T[i * self.p:(i + 1) * self.p, j * self.m:(j + 1) * self.m] = blk
        return np.linalg.pinv(O) @ (Y - T @ U)

    def equilibrium_input_from_output(self, y_eq):
        """utilities/initial_state_estimation.py:171-204."""
        M = self.C @ np.linalg.inv(np.eye(self.ns) - self.A) @ self.B + self.D
        return np.linalg.pinv(M) @ y_eq

    def equilibrium_state_from_input(self, u_eq):
        return np.linalg.inv(np.eye(self.ns) - self.A) @ self.B @ u_eq


def generate_instance(seed: int, N: int = 400, plant_params=None, u_range=(-1.0, 1.0)):
    """Reference RNG draw order for one controller instance:
    utilities/controller/controller_operation.py:59-75 (state randomisation),
    examples/direct_data_driven_mpc_example.py:282-287 (set_state(x_0)),
    controller_operation.py:126-133 (u_d, w_d, simulate N)."""
    pp = plant_params or FOUR_TANK
    plant = Plant(**pp)
    rng = np.random.default_rng(seed)
    ns, m, p, eps = plant.ns, plant.m, plant.p, plant.eps_max
    x_i0 = rng.uniform(-1.0, 1.0, size=ns)
    plant.x = x_i0
    u_i = rng.uniform(*u_range, (ns, m))
    w_i = eps * rng.uniform(-1.0, 1.0, (ns, p))
    y_i = plant.simulate(u_i, w_i)
    x_0 = plant.initial_state_from_trajectory(u_i.flatten(), y_i.flatten())
    plant.x = x_0
    u_d = rng.uniform(*u_range, (N, m))
    w_d = eps * rng.uniform(-1.0, 1.0, (N, p))
    y_d = plant.simulate(u_d, w_d)
    return dict(x_0=x_0, u_d=u_d, y_d=y_d, plant=plant, rng=rng)


def spec_from_params(params=None, **over) -> QPSpec:
    """Parameter derivation of utilities/controller/controller_creation.py:105-168
    (Q=q*I, R=r*I, lamb_alpha=lambda/eps_bar, c=1)."""
    d = dict(EXAMPLE_PARAMS)
    if params:
        d.update(params)
    d.update(over)
    m, p = len(d["u_s"]), len(d["y_s"])
    L, n = d["L"], d["n"]
    eps = d["epsilon_bar"]
    lamb_alpha = d["lambda_alpha_epsilon_bar"] / eps if eps != 0 else 1000.0
    slack = {0: "none", 1: "convex"}[d["slack_var_constraint_type"]]
    return QPSpec(n=n, m=m, p=p, L=L, Q=d["Q_scalar"] * np.eye(p * L),
                  R=d["R_scalar"] * np.eye(m * L), u_s=np.array(d["u_s"], float),
                  y_s=np.array(d["y_s"], float), robust=bool(d["controller_type"]),
                  eps_max=eps, lamb_alpha=lamb_alpha, lamb_sigma=d["lambda_sigma"],
                  c=1.0, slack=slack, tec=d.get("tec", True))


def closed_loop(spec: QPSpec, u_d, y_d, plant: Plant, w_sys: np.ndarray,
                n_mpc_step: int = 1, u_past=None, y_past=None):
    """Algorithm 1 / n-step Algorithm 2 driver,
    utilities/controller/controller_operation.py:259-305.  Returns (u_sys, y_sys)."""
    n, m, p = spec.n, spec.m, spec.p
    n_steps = w_sys.shape[0]
    up = (u_d[-n:].reshape(-1) if u_past is None else np.asarray(u_past, float).reshape(-1)).copy()
    yp = (y_d[-n:].reshape(-1) if y_past is None else np.asarray(y_past, float).reshape(-1)).copy()
    u_sys = np.zeros((n_steps, m))
    y_sys = np.zeros((n_steps, p))
    for t in range(0, n_steps, n_mpc_step):
        sol = solve_fullspace(spec, u_d, y_d, up, yp)
        if sol.status not in (OPTIMAL, OPTIMAL_INACCURATE):
            raise ValueError("MPC problem was not solved optimally.")
        for k in range(t, min(t + n_mpc_step, n_steps)):
            u = sol.optimal_u[(k - t) * m:(k - t + 1) * m]
            y = plant.step(u, w_sys[k])
            u_sys[k], y_sys[k] = u, y
            up = np.concatenate([up[m:], u])          # FIFO, controller.py:893-895
            yp = np.concatenate([yp[p:], y])
    return u_sys, y_sys
