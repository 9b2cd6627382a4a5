"""CPU tests: the compiled C restatement (oracle/ddmpc_oracle_c.c, the bench's cpu_baseline) against the
golden vectors and the numpy full-space oracle.  No GPU."""
import numpy as np
import pytest

from oracle import ddmpc_oracle as orc
from oracle import oracle_c

N, L, n, m, p = 400, 30, 4, 2, 2


def _batch(golden):
    u_d = np.stack([golden[f"s{s}_u_d"] for s in range(5)])
    y_d = np.stack([golden[f"s{s}_y_d"] for s in range(5)])
    return u_d, y_d, u_d[:, -n:, :].reshape(5, -1).copy(), y_d[:, -n:, :].reshape(5, -1).copy()


@pytest.mark.parametrize("tag,kw", [("none", {}), ("convex", dict(slack_var_constraint_type=1)), ("ucon", dict(tec=False))])
@pytest.mark.parametrize("structured", [True, False])
def test_c_restatement_matches_golden_solutions(golden, tag, kw, structured):
    u_d, y_d, up, yp = _batch(golden)
    spec = orc.spec_from_params(**kw)
    u, c, st, it = oracle_c.solve_batch(spec, N, u_d, y_d, up, yp, threads=2, structured=structured)
    assert np.all(st == 0)
    for s in range(5):
        ur, cr = golden[f"s{s}_{tag}_u"], float(golden[f"s{s}_{tag}_cost"][0])
        assert np.max(np.abs(u[s] - ur)) / np.max(np.abs(ur)) < 1e-9
        assert abs(c[s] - cr) / abs(cr) < 1e-10
    if tag == "convex":
        assert np.all(it >= 2)          # the box is active on the example data: at least one re-factorisation


def test_c_restatement_other_sizes_and_diagonal_weights():
    rng = np.random.default_rng(3)
    Ls, Ns = 10, 120
    q = rng.uniform(1.0, 4.0, p * Ls); r = rng.uniform(1e-4, 1e-2, m * Ls)
    spec = orc.spec_from_params(L=Ls, N=Ns, slack_var_constraint_type=1)
    spec.Q, spec.R = np.diag(q), np.diag(r)
    insts = [orc.generate_instance(s, N=Ns) for s in range(3)]
    u_d = np.stack([i["u_d"] for i in insts]); y_d = np.stack([i["y_d"] for i in insts])
    up = u_d[:, -n:, :].reshape(3, -1).copy(); yp = y_d[:, -n:, :].reshape(3, -1).copy()
    u, c, st, it = oracle_c.solve_batch(spec, Ns, u_d, y_d, up, yp)
    for b in range(3):
        sol = orc.solve_fullspace(spec, u_d[b], y_d[b], up[b], yp[b])
        assert st[b] == 0 and it[b] == max(sol.iters, 1)
        assert np.max(np.abs(u[b] - sol.optimal_u)) / np.max(np.abs(sol.optimal_u)) < 1e-9
        assert abs(c[b] - sol.cost) / abs(sol.cost) < 1e-10


def test_c_restatement_thread_count_does_not_change_results(golden):
    u_d, y_d, up, yp = _batch(golden)
    spec = orc.spec_from_params()
    a = oracle_c.solve_batch(spec, N, u_d, y_d, up, yp, threads=1)
    b = oracle_c.solve_batch(spec, N, u_d, y_d, up, yp, threads=4)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
