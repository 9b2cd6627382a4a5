"""CPU tests: the oracle against the reference-derived golden vectors, the
third-party cross-check, and its own second formulation.  No GPU."""
import numpy as np
import pytest

from oracle import ddmpc_oracle as orc
from oracle.reduced_form import solve_reduced

N, L, n, m, p = 400, 30, 4, 2, 2


def test_hankel_docstring_known_answer(golden):
    # direct_data_driven_mpc/utilities/hankel_matrix.py:26-37
    H = orc.hankel_matrix(golden["kat_X"], 2)
    assert H.shape == (4, 3)
    assert np.array_equal(H, golden["kat_H"])
    expect = np.array([[0.27392337, -0.91805295, 0.62654048],
                       [-0.46042657, -0.96694473, 0.82551115],
                       [-0.91805295, 0.62654048, 0.21327155],
                       [-0.96694473, 0.82551115, 0.45899312]])
    assert np.allclose(H, expect, atol=5e-9)


def test_hankel_matches_reference_on_generated_data(golden):
    wts = None
    for s in range(5):
        Hu = orc.hankel_matrix(golden[f"s{s}_u_d"], L + n)
        Hy = orc.hankel_matrix(golden[f"s{s}_y_d"], L + n)
        assert Hu.shape == (m * (L + n), N - L - n + 1)
        if wts is None:
            wts = np.cos(np.arange(Hu.size, dtype=float)).reshape(Hu.shape)
        for H, key in ((Hu, "Hu"), (Hy, "Hy")):
            fp = np.array([H.sum(), (H * wts).sum(), H[5, 7], H[-1, -1]])
            assert np.array_equal(fp, golden[f"s{s}_{key}_fp"])
    assert np.array_equal(orc.hankel_matrix(golden["s0_u_d"], L + n), golden["s0_Hu"])


def test_hankel_rejects_short_input():
    with pytest.raises(ValueError, match="N must be greater than or equal to L"):
        orc.hankel_matrix(np.zeros((3, 2)), 4)


def test_persistent_excitation_ranks(golden):
    for s in range(5):
        rank, ok = orc.evaluate_persistent_excitation(golden[f"s{s}_u_d"], L + 2 * n)
        assert [rank, int(ok)] == golden[f"s{s}_pe_rank"].tolist() == [76, 1]
    rank, ok = orc.evaluate_persistent_excitation(np.ones((N, m)), L + 2 * n)
    assert [rank, int(ok)] == golden["const_pe_rank"].tolist()
    assert not ok


def test_data_generation_matches_reference(golden):
    for s in range(5):
        inst = orc.generate_instance(s)
        assert np.array_equal(inst["u_d"], golden[f"s{s}_u_d"])
        assert np.array_equal(inst["y_d"], golden[f"s{s}_y_d"])
        assert np.array_equal(inst["x_0"], golden[f"s{s}_x0"])
    plant = orc.Plant(**orc.FOUR_TANK)
    assert np.allclose(plant.equilibrium_input_from_output(np.array([0.4, 0.4])), golden["eq_u"], atol=1e-14)


@pytest.mark.parametrize("tag,kw", [("none", {}), ("convex", dict(slack_var_constraint_type=1)),
                                    ("ucon", dict(tec=False))])
def test_oracle_solutions_pinned(golden, tag, kw):
    spec = orc.spec_from_params(**kw)
    for s in range(5):
        u_d, y_d = golden[f"s{s}_u_d"], golden[f"s{s}_y_d"]
        sol = orc.solve_fullspace(spec, u_d, y_d, u_d[-n:].reshape(-1), y_d[-n:].reshape(-1))
        assert sol.status == "optimal"
        assert np.allclose(sol.optimal_u, golden[f"s{s}_{tag}_u"], rtol=0, atol=1e-10)
        assert abs(sol.cost - golden[f"s{s}_{tag}_cost"][0]) < 1e-11


def test_survey_known_answers(golden):
    # SURVEY.md section 6 / BASELINE.md section 2 (seed 0, robust, slack NONE, TEC)
    spec = orc.spec_from_params()
    u_d, y_d = golden["s0_u_d"], golden["s0_y_d"]
    sol = orc.solve_fullspace(spec, u_d, y_d, u_d[-n:].reshape(-1), y_d[-n:].reshape(-1))
    assert np.allclose(sol.optimal_u[:2], [21.22188171, 20.30350327], atol=5e-9)
    assert abs(sol.cost - 4.543214030) < 5e-10
    assert abs(np.max(np.abs(sol.sigma[n * p:])) - 2.1368e-3) < 5e-8
    assert abs(np.sum(np.abs(sol.alpha)) - 53.34) < 5e-3


@pytest.mark.parametrize("tag", ["none", "convex"])
def test_third_party_solver_agrees(golden, tag):
    # scipy.optimize trust-constr on the same full-space QP (tests/golden/make_golden.py)
    u_ref, c_ref = golden[f"scipy_s0_{tag}_u"], golden[f"scipy_s0_{tag}_cost"][0]
    u, c = golden[f"s0_{tag}_u"], golden[f"s0_{tag}_cost"][0]
    assert abs(c - c_ref) / abs(c_ref) < 1e-9
    assert np.max(np.abs(u - u_ref)) / np.max(np.abs(u_ref)) < 1e-7


@pytest.mark.parametrize("kw", [{}, dict(slack_var_constraint_type=1), dict(tec=False),
                                dict(tec=False, slack_var_constraint_type=1)])
def test_kkt_certificate_and_second_formulation(kw):
    spec = orc.spec_from_params(**kw)
    for s in (0, 7):
        inst = orc.generate_instance(s)
        u_d, y_d = inst["u_d"], inst["y_d"]
        up, yp = u_d[-n:].reshape(-1), y_d[-n:].reshape(-1)
        sol = orc.solve_fullspace(spec, u_d, y_d, up, yp)
        cert = orc.kkt_certificate(spec, u_d, y_d, up, yp, sol.x)
        assert cert["res_eq"] < 1e-11 and cert["res_box"] < 1e-15
        assert cert["res_stat"] < 1e-10 * max(1.0, cert["grad_scale"]) and cert["dual_sign"] == 0.0
        red = solve_reduced(spec, u_d, y_d, up, yp)
        assert red["status"] == "optimal"
        assert np.max(np.abs(red["optimal_u"] - sol.optimal_u)) / np.max(np.abs(sol.optimal_u)) < 1e-10
        assert abs(red["cost"] - sol.cost) / abs(sol.cost) < 1e-10
        assert np.max(np.abs(red["sigma"] - sol.sigma)) < 1e-11
        if spec.slack == "convex":
            assert np.array_equal(red["act"], sol.active)


def test_nominal_full_row_rank_known_answer():
    # noisy Hankel has full row rank => optimal_u == tile(u_s, L), cost 0
    spec = orc.spec_from_params(controller_type=0)
    inst = orc.generate_instance(0)
    u_d, y_d = inst["u_d"], inst["y_d"]
    sol = orc.solve_fullspace(spec, u_d, y_d, u_d[-n:].reshape(-1), y_d[-n:].reshape(-1))
    assert np.allclose(sol.optimal_u, np.tile(spec.u_s, L), atol=1e-7)
    assert abs(sol.cost) < 1e-9
    red = solve_reduced(spec, u_d, y_d, u_d[-n:].reshape(-1), y_d[-n:].reshape(-1))
    assert np.allclose(red["optimal_u"], np.tile(spec.u_s, L), atol=1e-12)


def test_closed_loop_converges_to_setpoint():
    # seed 0 example closes at y ~ (0.65, 0.77), u ~ (1, 1); n-step scheme n_mpc_step = n
    spec = orc.spec_from_params()
    inst = orc.generate_instance(0)
    n_steps = 161
    w = inst["plant"].eps_max * inst["rng"].uniform(-1.0, 1.0, (n_steps, p))
    u_sys, y_sys = orc.closed_loop(spec, inst["u_d"], inst["y_d"], inst["plant"], w, n_mpc_step=4)
    assert np.allclose(u_sys[0], [21.22188171, 20.30350327], atol=5e-9)
    assert np.all(np.abs(y_sys[-1] - spec.y_s) < 0.03)
    assert np.all(np.abs(u_sys[-1] - spec.u_s) < 0.2)


def test_nominal_exact_oracle():
    # oracle/nominal_exact.py (SVD route for rank-deficient data) against the cases with an independent answer:
    # full-row-rank noisy data -> u = u_s, cost 0; exact data + rounded setpoint -> infeasible, and its least-squares
    # compromise is what the full-space minimum-norm KKT solve returns; exact data + equilibrium setpoint -> optimal
    from oracle.nominal_exact import solve_nominal_exact
    from direct_data_driven_mpc_amd.harness import FOUR_TANK, generate_batch
    spec = orc.spec_from_params(controller_type=0)
    inst = orc.generate_instance(0)
    up = inst["u_d"][-4:].reshape(-1); yp = inst["y_d"][-4:].reshape(-1)
    r = solve_nominal_exact(spec, inst["u_d"], inst["y_d"], up, yp)
    assert r["status"] == "optimal" and r["rank"] == 136
    assert np.max(np.abs(r["optimal_u"] - np.tile(spec.u_s, spec.L))) < 1e-9 and abs(r["cost"]) < 1e-12
    plant = dict(FOUR_TANK); plant["eps_max"] = 0.0
    d = generate_batch([0], N=400, plant=plant)
    up = d["u_d"][0, -4:].reshape(-1); yp = d["y_d"][0, -4:].reshape(-1)
    r = solve_nominal_exact(spec, d["u_d"][0], d["y_d"][0], up, yp)
    assert r["status"] == "infeasible" and r["rank"] == 72 and 1e-6 < r["residual"] < 1e-3
    full = orc.solve_fullspace(spec, d["u_d"][0], d["y_d"][0], up, yp)
    assert np.max(np.abs(r["optimal_u"] - full.optimal_u)) / np.max(np.abs(full.optimal_u)) < 1e-6
    A, Bm, Cm, D = (FOUR_TANK[k] for k in "ABCD")
    spec.y_s = (Cm @ np.linalg.inv(np.eye(4) - A) @ Bm + D) @ spec.u_s
    r = solve_nominal_exact(spec, d["u_d"][0], d["y_d"][0], up, yp)
    assert r["status"] == "optimal" and r["residual"] < 1e-12 and r["cost"] > 0.1


def test_nominal_oracles_against_the_extended_precision_golden_solutions():
    # tests/golden/cfg5_extended.npz: BASELINE configs[4] solved from the DATA alone in 80-bit arithmetic with orthogonal
    # factorisations (make_golden_cfg5.py; the reference's formulation, controller.py:506-538,679-711).  The fp64 checkers the GPU
    # tests use on exact data -- the SVD route and the model-based solution -- must sit inside the parity bar of it.
    import os
    from direct_data_driven_mpc_amd.harness import generate_batch
    from oracle.nominal_exact import solve_nominal_exact, solve_nominal_model_based
    sys_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    z = np.load(os.path.join(sys_path, "cfg5_extended.npz"))
    import importlib.util
    spec_ = importlib.util.spec_from_file_location("make_golden_cfg5", os.path.join(sys_path, "make_golden_cfg5.py"))
    g = importlib.util.module_from_spec(spec_); spec_.loader.exec_module(g)
    spec, plant, N = g.config5(512)
    n = spec.n
    for k, b in list(enumerate(z["instances"]))[:3]:                 # (283 first; a few seconds each)
        d = generate_batch([int(b)], N=N, plant=plant)
        up = d["u_d"][0, -n:, :].reshape(-1); yp = d["y_d"][0, -n:, :].reshape(-1)
        sc = np.max(np.abs(z["optimal_u"][k]))
        mod = solve_nominal_model_based(spec, plant, up, yp)
        assert np.max(np.abs(mod["optimal_u"] - z["optimal_u"][k])) < 1e-8 * sc and abs(mod["cost"] - z["cost"][k]) < 1e-9 * z["cost"][k]
        svd = solve_nominal_exact(spec, d["u_d"][0], d["y_d"][0], up, yp)
        assert svd["status"] == "optimal" and np.max(np.abs(svd["optimal_u"] - z["optimal_u"][k])) < 2e-8 * sc


@pytest.mark.parametrize("cfg", ["cfg2", "cfg4"])
@pytest.mark.parametrize("slack", ["none", "convex"])
def test_fp64_checkers_against_the_extended_precision_golden_solutions_of_the_headline_config(slack, cfg):
    # tests/golden/cfg2_extended.npz: BASELINE configs[1] (four-tank robust, L = 30, N = 400), the QP as the reference states it
    # (full-space KKT of build_fullspace_qp), solved in 80-bit arithmetic with the active set verified there
    # (make_golden_cfg2_extended.py).  The fp64 checkers the GPU tests use -- the numpy full-space oracle, its reduced form and
    # the compiled C restatement -- sit within 1e-12 of it; the active-set iteration counts agree.
    import os
    from direct_data_driven_mpc_amd.harness import generate_batch
    # (cfg4: BASELINE configs[3], L = 60, N = 1000, two instances)
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", cfg + "_extended.npz"))
    seeds = [int(s) for s in z["seeds"]]
    Lh, N = int(z["L"]), int(z["N"])
    spec = orc.spec_from_params(L=Lh, N=N, **({"slack_var_constraint_type": 1} if slack == "convex" else {}))
    n = spec.n
    d = generate_batch(seeds, N=N)
    up = d["u_d"][:, -n:, :].reshape(len(seeds), -1).copy(); yp = d["y_d"][:, -n:, :].reshape(len(seeds), -1).copy()
    ug, cg = z["optimal_u_" + slack], z["cost_" + slack]
    for k in range(len(seeds)):
        sol = orc.solve_fullspace(spec, d["u_d"][k], d["y_d"][k], up[k], yp[k])
        assert sol.status == "optimal" and np.max(np.abs(sol.optimal_u - ug[k])) < 1e-12 * np.max(np.abs(ug[k])) and abs(sol.cost - cg[k]) < 1e-12 * cg[k]
        assert np.max(np.abs(sol.alpha - z["alpha_" + slack][k])) < 1e-10 * np.max(np.abs(z["alpha_" + slack][k]))
        if slack == "convex":
            assert sol.iters == int(z["iters_convex"][k]) and int(np.count_nonzero(sol.active)) == int(z["n_active_convex"][k])
        red = solve_reduced(spec, d["u_d"][k], d["y_d"][k], up[k], yp[k])
        assert np.max(np.abs(red["optimal_u"] - ug[k])) < 5e-11 * np.max(np.abs(ug[k])) and abs(red["cost"] - cg[k]) < 5e-11 * cg[k]
    try:
        from oracle import oracle_c
        oracle_c.load()
    except Exception:
        pytest.skip("the compiled C restatement is not built")
    u_c, c_c, st_c, it_c = oracle_c.solve_batch(spec, N, d["u_d"], d["y_d"], up, yp)[:4]
    assert np.all(st_c == 0)
    assert np.max(np.abs(u_c - ug) / np.max(np.abs(ug), axis=1, keepdims=True)) < 5e-11 and np.max(np.abs(c_c - cg) / cg) < 5e-11


def test_slack_box_as_rank_k_update_of_the_first_factor():
    """The algebra behind round 5's CONVEX path (ddmpc_cold2.hpp `rank_update`, ddmpc_rr3.hpp), on the CPU: active-set iterations
    that keep the Cholesky factor of the EMPTY active set and treat the switched slack components as a rank-k diagonal
    modification (Woodbury) visit the same active sets in the same number of iterations as re-factoring, and end in the same
    solution -- which is the full-space oracle's (controller.py:631-677).  k stays tiny on the benchmark data."""
    import scipy.linalg as sla
    from oracle.reduced_form import component_tables
    spec = orc.spec_from_params(slack_var_constraint_type=1)
    n, m, p, L_, Ln = spec.n, spec.m, spec.p, spec.L, spec.Ln
    lam = spec.lamb_alpha * spec.eps_max
    dd = lam / spec.lamb_sigma
    bound = spec.c * spec.eps_max
    w_pred = slice(Ln * m + n * p, Ln * (m + p))
    for seed in range(6):
        inst = orc.generate_instance(seed)
        u_d, y_d = inst["u_d"], inst["y_d"]
        up, yp = u_d[-n:].reshape(-1), y_d[-n:].reshape(-1)
        H = np.vstack([orc.hankel_matrix(u_d, Ln), orc.hankel_matrix(y_d, Ln)])
        G = H @ H.T
        act = np.zeros(L_ * p, dtype=int)
        D0, t0 = component_tables(spec, up, yp, act)
        L0 = np.linalg.cholesky(G + lam * np.diag(D0))
        y0 = sla.solve_triangular(L0, t0, lower=True)
        iters, kmax = 0, 0
        while True:
            iters += 1
            S = np.nonzero(act)[0]
            k = len(S)
            kmax = max(kmax, k)
            v = y0
            if k:
                rows = np.arange(Ln * m + n * p, Ln * (m + p))[S]
                E = np.zeros((G.shape[0], k)); E[rows, np.arange(k)] = 1.0
                W = sla.solve_triangular(L0, E, lower=True)
                yq = y0 + bound * (W @ act[S])
                Sm = np.eye(k) / dd - W.T @ W
                assert np.all(np.linalg.eigvalsh(Sm) > 0)              # positive definite: K(act) is
                v = yq + W @ np.linalg.solve(Sm, W.T @ yq)
            beta = sla.solve_triangular(L0.T, v, lower=False)
            sh = -lam * beta[w_pred] / spec.lamb_sigma
            new = np.where(sh > bound, 1, np.where(sh < -bound, -1, 0))
            if np.array_equal(new, act):
                break
            act = new
        sol = orc.solve_fullspace(spec, u_d, y_d, up, yp)
        assert iters == sol.iters and kmax <= 4
        D, t = component_tables(spec, up, yp, act)
        z = t - lam * D * beta
        assert np.max(np.abs(z[n * m:Ln * m] - sol.optimal_u)) <= 1e-10 * np.max(np.abs(sol.optimal_u))


def test_model_based_nominal_oracle_with_dense_weighting_matrices():
    """controller.py:708-710 takes any PSD Q, R.  The model-based restatement of the NOMINAL QP on exact data (oracle/nominal_exact.py:
    the yardstick of the kernels beyond 271 rows) with dense matrices: against the full-space KKT oracle on a small exact-data case,
    and its dense code path against the diagonal one on diagonal matrices."""
    from oracle.nominal_exact import solve_nominal_model_based
    from direct_data_driven_mpc_amd.harness import generate_batch
    rng = np.random.default_rng(3)
    n, m, p, Lh = 2, 2, 2, 10
    A = rng.normal(size=(n, n)); A *= 0.8 / max(abs(np.linalg.eigvals(A)))
    plant = dict(A=A, B=rng.normal(size=(n, m)), C=rng.normal(size=(p, n)), D=np.zeros((p, m)), eps_max=0.0)
    u_s = 0.1 * np.ones(m); y_s = (plant["C"] @ np.linalg.inv(np.eye(n) - A) @ plant["B"]) @ u_s

    def spd(k, s):
        X = rng.normal(size=(k, k))
        return s * (np.eye(k) + 0.1 * (X @ X.T) / k)
    Q, R = spd(p * Lh, 3.0), spd(m * Lh, 0.05)
    mk = lambda Q_, R_: orc.QPSpec(n=n, m=m, p=p, L=Lh, Q=Q_, R=R_, u_s=u_s, y_s=y_s, robust=False, eps_max=0.0, lamb_alpha=0.0,
                                   lamb_sigma=0.0, c=0.0, slack="none", tec=True)
    d = generate_batch(range(1), N=120, plant=plant)
    up = d["u_d"][0, -n:, :].reshape(-1); yp = d["y_d"][0, -n:, :].reshape(-1)
    mod = solve_nominal_model_based(mk(Q, R), plant, up, yp)
    fs = orc.solve_fullspace(mk(Q, R), d["u_d"][0], d["y_d"][0], up, yp)
    assert fs.status == "optimal" and mod["feas_residual"] < 1e-10
    assert np.max(np.abs(mod["optimal_u"] - fs.optimal_u)) / np.max(np.abs(fs.optimal_u)) < 1e-9
    assert abs(mod["cost"] - fs.cost) / abs(fs.cost) < 1e-9
    Qd, Rd = np.diag(np.diag(Q)), np.diag(np.diag(R))
    a = solve_nominal_model_based(mk(Qd, Rd), plant, up, yp)
    b = solve_nominal_model_based(mk(Qd + 1e-300, Rd), plant, up, yp)       # (not diagonal to the letter: the dense path)
    assert np.max(np.abs(a["optimal_u"] - b["optimal_u"])) < 1e-12 and abs(a["cost"] - b["cost"]) < 1e-12 * max(1.0, abs(a["cost"]))
