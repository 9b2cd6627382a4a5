"""GPU tests added in round 4: API sequences the advisor found untested, and the restructured path for problems beyond the
register-resident kernels (cfg 5 at its full batch, channel counts other than four on the structured Gram, ...)."""
import os
import sys

import numpy as np
import pytest

from direct_data_driven_mpc_amd import _lib as L
from direct_data_driven_mpc_amd import harness
from direct_data_driven_mpc_amd.engine import BatchedDDMPC
from oracle import ddmpc_oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
TOL_U, TOL_COST = 1e-8, 1e-9


def _spec_engine(spec, N, B, **kw):
    return BatchedDDMPC(n=spec.n, m=spec.m, p=spec.p, L_=spec.L, N=N, Q=spec.Q, R=spec.R, u_s=spec.u_s, y_s=spec.y_s, batch=B,
                        controller_type=L.ROBUST if spec.robust else L.NOMINAL,
                        slack_type=L.SLACK_CONVEX if spec.slack == "convex" else L.SLACK_NONE, eps_max=spec.eps_max,
                        lamb_alpha=spec.lamb_alpha, lamb_sigma=spec.lamb_sigma, c=spec.c, use_terminal_constraint=spec.tec, **kw)


# ------------------------------------------------------------------ ddmpc_solve -> ddmpc_prepare -> ddmpc_get_solution
def test_prepare_after_solve_keeps_the_solution_readable(gpu):
    # NOMINAL controller beyond the register-resident kernels (the cfg-5 shape): ddmpc_prepare re-forms the data-dependent
    # factors and writes no solution; what the solve before it left for ddmpc_get_solution (z, x = L^-T w, the flags) must
    # stay readable and unchanged (controller.py:434-438: the `.value` stand-ins)
    from test_gpu_round3 import _config5
    B = 3
    spec, plant, N, d, up, yp = _config5(B)
    with _spec_engine(spec, N, B) as eng:
        eng.set_data(d["u_d"], d["y_d"])
        u, cost, status, _ = (x.copy() for x in eng.solve(up, yp))
        ub0, yb0, al0 = eng.get_solution("ubar"), eng.get_solution("ybar"), eng.get_solution("alpha")
        eng.prepare()
        ub1, yb1, al1 = eng.get_solution("ubar"), eng.get_solution("ybar"), eng.get_solution("alpha")
        w = tuple(x.copy() for x in eng.step(up, yp))
    assert np.all(status == 0)
    assert np.array_equal(ub0, ub1) and np.array_equal(yb0, yb1) and np.array_equal(al0, al1)
    assert np.array_equal(ub1[:, spec.n * spec.m:], u) and np.all(np.isfinite(al1))
    assert np.array_equal(w[0], u) and np.array_equal(w[1], cost)


# ------------------------------------------------------------------ AUTO refinement of the affine law, zero setpoints
def test_auto_refinement_of_the_affine_law_with_zero_setpoints(gpu):
    # plain regulation (u_s = y_s = 0) on the ill-conditioned random plant of the sweep (case 1017): the factor-export solve
    # of ddmpc_prepare runs at the zero past window, where the right-hand side vanishes altogether -- its residual check can
    # flag nothing.  ddmpc_prepare therefore also probes every data set with a solve at the window a controller starts
    # from; the warm step must then meet the oracle at the standard bars under the default AUTO mode, as it does with
    # ALWAYS, and visibly better than with refinement OFF.
    import test_gpu_parity as T
    m, p, ns = 2, 3, 4
    plant = T._random_plant(np.random.default_rng(1017), ns, m, p, 0.002)
    Lh, N, B = 16, 200, 6
    spec = orc.QPSpec(n=ns, m=m, p=p, L=Lh, Q=2.0 * np.eye(p * Lh), R=0.05 * np.eye(m * Lh), u_s=np.zeros(m), y_s=np.zeros(p),
                      robust=True, eps_max=0.002, lamb_alpha=20.0, lamb_sigma=500.0, c=1.0, slack="none", tec=True)
    d = harness.generate_batch(range(170, 170 + B), N=N, plant=plant)
    up = d["u_d"][:, 60:60 + ns, :].reshape(B, -1).copy(); yp = d["y_d"][:, 60:60 + ns, :].reshape(B, -1).copy()
    warm, gain = {}, {}
    with _spec_engine(spec, N, B) as eng:
        eng.set_data(d["u_d"], d["y_d"])
        for mode in ("off", "auto", "always"):
            eng.set_refinement(mode)                   # (also forgets the affine law)
            warm[mode] = tuple(x.copy() for x in eng.step(up, yp))
            gain[mode] = eng.gain()
    err = {}
    for mode, (u, c, s, _) in warm.items():
        eu = ec = 0.0
        for b in range(B):
            sol = orc.solve_fullspace(spec, d["u_d"][b], d["y_d"][b], up[b], yp[b])
            assert sol.status == "optimal" and s[b] == 0
            eu = max(eu, np.max(np.abs(u[b] - sol.optimal_u)) / np.max(np.abs(sol.optimal_u)))
            ec = max(ec, abs(c[b] - sol.cost) / abs(sol.cost))
        err[mode] = (eu, ec)
    assert err["always"][0] < TOL_U and err["always"][1] < TOL_COST, err
    assert err["auto"][0] < TOL_U and err["auto"][1] < TOL_COST, err
    assert err["off"][0] > 10 * err["auto"][0], err          # the unrefined law is visibly worse: the probe did flag
    # the law itself: AUTO's equals ALWAYS's where it was refined (here: every instance)
    assert np.max(np.abs(gain["auto"] - gain["always"])) <= 1e-9 * np.max(np.abs(gain["always"]))
