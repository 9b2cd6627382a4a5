"""Generate the golden fixtures under tests/golden/.

Run ONCE in the build container (the only place /root/reference exists):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What comes from the reference itself (imported, never copied):
  * `hankel_matrix`, `evaluate_persistent_excitation`
    (direct_data_driven_mpc/utilities/hankel_matrix.py) -- the docstring
    known-answer and the Hankel matrices / PE ranks of the generated data;
  * `LTISystemModel` (utilities/model_simulation.py) loaded from the reference's
    own four-tank YAML -- the trajectories u_d, y_d, x_0 for seeds 0..4 drawn in
    the RNG order of utilities/controller/controller_operation.py:59-75,126-133
    and examples/direct_data_driven_mpc_example.py:282-287, plus the equilibrium
    pair used by examples/robust_data_driven_mpc_reproduction.py:223-228.
The reference's QP solve cannot run here (cvxpy is not installed), so the QP
solutions stored below come from oracle/ddmpc_oracle.py and, for seed 0, from a
third-party solver (scipy.optimize trust-constr) on the same full-space problem.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(1, REF)

from direct_data_driven_mpc.utilities.hankel_matrix import (  # noqa: E402  (reference)
    hankel_matrix as ref_hankel, evaluate_persistent_excitation as ref_pe)
from utilities.model_simulation import LTISystemModel  # noqa: E402  (reference)

from oracle import ddmpc_oracle as orc  # noqa: E402

N, L, n, m, p = 400, 30, 4, 2, 2


def reference_instance(seed):
    model = LTISystemModel(
        config_file=os.path.join(REF, "examples/config/models/four_tank_system_params.yaml"),
        model_key_value="FourTankSystem")
    rng = np.random.default_rng(seed)
    ns = model.get_system_order()
    eps = model.get_eps_max()
    x_i0 = rng.uniform(-1.0, 1.0, size=ns)
    model.set_state(state=x_i0)
    u_i = rng.uniform(-1.0, 1.0, (ns, m))
    w_i = eps * rng.uniform(-1.0, 1.0, (ns, p))
    y_i = model.simulate(U=u_i, W=w_i, steps=ns)
    x_0 = model.get_initial_state_from_trajectory(U=u_i.flatten(), Y=y_i.flatten())
    model.set_state(state=x_0)
    u_d = rng.uniform(-1.0, 1.0, (N, m))
    w_d = eps * rng.uniform(-1.0, 1.0, (N, p))
    y_d = model.simulate(U=u_d, W=w_d, steps=N)
    return model, x_0, u_d, y_d


def main():
    out = {}
    # 1. docstring known answer, hankel_matrix.py:26-37
    Xk = np.random.default_rng(0).uniform(-1, 1, (4, 2))
    out["kat_X"] = Xk
    out["kat_H"] = ref_hankel(Xk, 2)
    # 2. reference-generated trajectories + Hankel fingerprints
    wts = None
    for seed in range(5):
        model, x_0, u_d, y_d = reference_instance(seed)
        out[f"s{seed}_x0"] = x_0
        out[f"s{seed}_u_d"] = u_d
        out[f"s{seed}_y_d"] = y_d
        Hu = ref_hankel(u_d, L + n)
        Hy = ref_hankel(y_d, L + n)
        if wts is None:
            wts = np.cos(np.arange(Hu.size, dtype=float)).reshape(Hu.shape)
        out[f"s{seed}_Hu_fp"] = np.array([Hu.sum(), (Hu * wts).sum(), Hu[5, 7], Hu[-1, -1]])
        out[f"s{seed}_Hy_fp"] = np.array([Hy.sum(), (Hy * wts).sum(), Hy[5, 7], Hy[-1, -1]])
        rank, ok = ref_pe(u_d, L + 2 * n)
        out[f"s{seed}_pe_rank"] = np.array([rank, int(ok)])
        if seed == 0:
            out["s0_Hu"] = Hu
    # a deliberately non-exciting input must fail the rank test
    rank, ok = ref_pe(np.ones((N, m)), L + 2 * n)
    out["const_pe_rank"] = np.array([rank, int(ok)])
    # 3. equilibrium pair for y_0 = [0.4, 0.4] (reproduction script defaults)
    y0 = np.array([0.4, 0.4]).reshape(-1, 1)
    u_eq = model.get_equilibrium_input_from_output(y_eq=y0)
    out["eq_u"] = np.asarray(u_eq).reshape(-1)
    # 4. oracle QP solutions (regression pins) for seeds 0..4, three variants
    for seed in range(5):
        u_d, y_d = out[f"s{seed}_u_d"], out[f"s{seed}_y_d"]
        up, yp = u_d[-n:].reshape(-1), y_d[-n:].reshape(-1)
        for tag, kw in (("none", {}), ("convex", dict(slack_var_constraint_type=1)),
                        ("ucon", dict(tec=False))):
            spec = orc.spec_from_params(**kw)
            sol = orc.solve_fullspace(spec, u_d, y_d, up, yp)
            assert sol.status == "optimal"
            out[f"s{seed}_{tag}_u"] = sol.optimal_u
            out[f"s{seed}_{tag}_cost"] = np.array([sol.cost])
    # 5. third-party cross-check (scipy trust-constr) on seed 0
    from scipy.optimize import Bounds, LinearConstraint, minimize
    u_d, y_d = out["s0_u_d"], out["s0_y_d"]
    up, yp = u_d[-n:].reshape(-1), y_d[-n:].reshape(-1)
    for tag, kw in (("none", {}), ("convex", dict(slack_var_constraint_type=1))):
        spec = orc.spec_from_params(**kw)
        qp = orc.build_fullspace_qp(spec, u_d, y_d, up, yp)
        nx = qp.P.shape[0]
        lb = np.full(nx, -np.inf)
        ub = np.full(nx, np.inf)
        lb[qp.box_idx] = -qp.bound
        ub[qp.box_idx] = qp.bound
        res = minimize(lambda x: x @ qp.P @ x + qp.q @ x + qp.const, np.zeros(nx),
                       jac=lambda x: 2 * qp.P @ x + qp.q, hess=lambda x: 2 * qp.P,
                       method="trust-constr",
                       constraints=[LinearConstraint(qp.A, qp.b, qp.b)],
                       bounds=Bounds(lb, ub) if qp.box_idx.size else None,
                       options=dict(gtol=1e-12, xtol=1e-14, barrier_tol=1e-12, maxiter=3000))
        out[f"scipy_s0_{tag}_u"] = res.x[qp.sl["ubar"]][n * m:]
        out[f"scipy_s0_{tag}_cost"] = np.array([res.fun])
    np.savez_compressed(os.path.join(HERE, "four_tank_golden.npz"), **out)
    print("wrote", os.path.join(HERE, "four_tank_golden.npz"), len(out), "arrays")


if __name__ == "__main__":
    main()
