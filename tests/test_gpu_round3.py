"""GPU tests added in round 3: BASELINE configs[3] and configs[4] at (or near) their stated batches inside the suite the
driver runs, the default refinement mode on the data it exists for, and shapes at the edges of the residual check."""
import os
import sys

import numpy as np
import pytest

from direct_data_driven_mpc_amd import _lib as L
from direct_data_driven_mpc_amd import harness
from direct_data_driven_mpc_amd.engine import BatchedDDMPC
from oracle import ddmpc_oracle as orc
from oracle import oracle_c

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
TOL_U, TOL_COST = 1e-8, 1e-9


def _spec_engine(spec, N, B, **kw):
    return BatchedDDMPC(n=spec.n, m=spec.m, p=spec.p, L_=spec.L, N=N, Q=spec.Q, R=spec.R, u_s=spec.u_s, y_s=spec.y_s, batch=B,
                        controller_type=L.ROBUST if spec.robust else L.NOMINAL,
                        slack_type=L.SLACK_CONVEX if spec.slack == "convex" else L.SLACK_NONE, eps_max=spec.eps_max,
                        lamb_alpha=spec.lamb_alpha, lamb_sigma=spec.lamb_sigma, c=spec.c, use_terminal_constraint=spec.tec, **kw)


def _threads():
    import bench
    return bench.host_cores()


# ------------------------------------------------------------------ cfg 4 at its stated batch
@pytest.mark.parametrize("slack", [0, 1], ids=["none", "convex"])
def test_config4_at_its_stated_batch(gpu, slack):
    # BASELINE configs[3]: L = 60, N = 1000 robust scheme, batch = 1024 on one GPU (the <17,8> kernel instance), every instance
    # against the compiled CPU restatement, a sample against the full-space oracle (the reference's own formulation), default
    # refinement mode; slack CONVEX: the same active-set iteration counts
    import torch
    B, Lh, N = 1024, 60, 1000
    spec = orc.spec_from_params(L=Lh, N=N, slack_var_constraint_type=slack)
    d = harness.generate_batch(range(B), N=N)
    n = spec.n
    up = d["u_d"][:, -n:, :].reshape(B, -1).copy(); yp = d["y_d"][:, -n:, :].reshape(B, -1).copy()
    u_ref, c_ref, st_ref, it_ref = oracle_c.solve_batch(spec, N, d["u_d"], d["y_d"], up, yp, threads=_threads())
    assert not np.count_nonzero(st_ref)
    dev = torch.device("cuda", 0)
    t = lambda x: torch.from_numpy(x).to(dev)
    with _spec_engine(spec, N, B) as eng:
        assert "17,8" in eng.kernel_name()
        eng.set_data(t(d["u_d"]), t(d["y_d"]))
        out = eng.solve(t(up), t(yp)); torch.cuda.synchronize()
        u, c, st, it = (x.cpu().numpy() for x in out)
    assert not np.count_nonzero(st)
    eu = np.max(np.max(np.abs(u - u_ref), axis=1) / np.max(np.abs(u_ref), axis=1))
    ec = np.max(np.abs(c - c_ref) / np.abs(c_ref))
    assert eu < TOL_U and ec < TOL_COST, (eu, ec)
    if slack:
        assert np.array_equal(it, it_ref) and it.max() >= 2
    for b in (0, 511, 1023):
        sol = orc.solve_fullspace(spec, d["u_d"][b], d["y_d"][b], up[b], yp[b])
        assert sol.status == "optimal"
        assert np.max(np.abs(u[b] - sol.optimal_u)) / np.max(np.abs(sol.optimal_u)) < TOL_U
        assert abs(c[b] - sol.cost) <= TOL_COST * abs(sol.cost)


# ------------------------------------------------------------------ cfg 5 beyond a handful of instances
def _config5(B):
    rng = np.random.default_rng(0)
    ns = n = 8; m = p = 8; Lh = 30; N = 2000
    A = rng.normal(size=(ns, ns)); A *= 0.9 / max(abs(np.linalg.eigvals(A)))
    plant = dict(A=A, B=rng.normal(size=(ns, m)), C=rng.normal(size=(p, ns)), D=np.zeros((p, m)), eps_max=0.0)
    u_s = 0.1 * np.ones(m)
    y_s = (plant["C"] @ np.linalg.inv(np.eye(ns) - A) @ plant["B"]) @ u_s
    spec = orc.QPSpec(n=n, m=m, p=p, L=Lh, Q=3.0 * np.eye(p * Lh), R=1e-4 * np.eye(m * Lh), u_s=u_s, y_s=y_s, robust=False,
                      eps_max=0.0, lamb_alpha=0.0, lamb_sigma=0.0, c=0.0, slack="none", tec=True)
    d = harness.generate_batch(range(B), N=N, plant=plant)
    up = d["u_d"][:, -n:, :].reshape(B, -1).copy(); yp = d["y_d"][:, -n:, :].reshape(B, -1).copy()
    return spec, plant, N, d, up, yp


def test_config5_at_a_quarter_of_its_stated_batch(gpu):
    # BASELINE configs[4] (nominal, m = p = 8, n = 8, L = 30, N = 2000, exact data; r = 608 rows, rank 312), 128 of its 512
    # instances, every one against the MODEL-BASED solution of the same QP (trajectory space from (A, B, C): well
    # conditioned, cheap) at the standard bars; alpha reconstructed: H alpha must reproduce [ubar; ybar]
    from oracle.nominal_exact import solve_nominal_model_based
    B = 128
    spec, plant, N, d, up, yp = _config5(B)
    with _spec_engine(spec, N, B) as eng:
        eng.set_data(d["u_d"], d["y_d"])
        u, cost, status, _ = eng.solve(up, yp)
        ub = eng.get_solution("ubar")
        yb = eng.get_solution("ybar")
        al = eng.get_solution("alpha")
    assert np.all(status == 0)
    n, m = spec.n, spec.m
    assert np.array_equal(ub[:, n * m:], u)
    worst_u = worst_c = 0.0
    for b in range(B):
        mod = solve_nominal_model_based(spec, plant, up[b], yp[b])
        sc = np.max(np.abs(mod["optimal_u"]))
        assert mod["feas_residual"] < 1e-10
        worst_u = max(worst_u, np.max(np.abs(u[b] - mod["optimal_u"])) / sc)
        worst_c = max(worst_c, abs(cost[b] - mod["cost"]) / abs(mod["cost"]))
    assert worst_u < TOL_U and worst_c < TOL_COST, (worst_u, worst_c)
    # `.alpha.value` (controller.py:434) on the rank-revealing route: a (minimum-norm) alpha with H alpha = [ubar; ybar]
    assert al.shape == (B, N - spec.Ln + 1) and np.all(np.isfinite(al))
    for b in (0, 63, 127):
        Hu, Hy = orc.hankel_matrix(d["u_d"][b], spec.Ln), orc.hankel_matrix(d["y_d"][b], spec.Ln)
        zu, zy = Hu @ al[b], Hy @ al[b]
        sc = max(np.max(np.abs(ub[b])), np.max(np.abs(yb[b])))
        assert np.max(np.abs(zu - ub[b])) < 1e-7 * sc and np.max(np.abs(zy - yb[b])) < 1e-7 * sc


# ------------------------------------------------------------------ the residual check at the edges of what it can stage
def test_auto_refinement_where_the_residual_check_does_not_fit(gpu):
    # a short horizon with a long trajectory: the smallest kernel instance (NT = 2) cannot stage alpha (c ~ 1900 entries)
    # in its LDS scratch, so an instance whose a-priori bound does not decide is refined unconditionally.  With a threshold
    # so low that the bound never decides, AUTO must equal refinement ALWAYS bit for bit; with the default threshold every
    # instance equals either its OFF or its ALWAYS result and meets the parity bars.
    rng = np.random.default_rng(5)
    m = p = 1; ns = n = 2; Lh = 6; N = 1900
    import test_gpu_parity as T
    plant = T._random_plant(rng, ns, m, p, 0.002)
    spec = orc.QPSpec(n=n, m=m, p=p, L=Lh, Q=2.0 * np.eye(p * Lh), R=0.05 * np.eye(m * Lh), u_s=np.array([0.2]), y_s=np.array([-0.1]),
                      robust=True, eps_max=0.002, lamb_alpha=20.0, lamb_sigma=500.0, c=1.0, slack="none", tec=True)
    B = 5
    d = harness.generate_batch(range(40, 40 + B), N=N, plant=plant)
    up = d["u_d"][:, -n:, :].reshape(B, -1).copy(); yp = d["y_d"][:, -n:, :].reshape(B, -1).copy()
    res = {}
    with _spec_engine(spec, N, B) as eng:
        assert "<2,1>" in eng.kernel_name()
        eng.set_data(d["u_d"], d["y_d"])
        for tag, mode, thr in (("off", "off", None), ("auto", "auto", -10.7), ("auto_low", "auto", -14.0), ("always", "always", None)):
            eng.set_refinement(mode, res_log10=thr)
            res[tag] = tuple(x.copy() for x in eng.solve(up, yp))
    assert np.array_equal(res["auto_low"][0], res["always"][0]) and np.array_equal(res["auto_low"][1], res["always"][1])
    for b in range(B):
        assert np.array_equal(res["auto"][0][b], res["off"][0][b]) or np.array_equal(res["auto"][0][b], res["always"][0][b])
        sol = orc.solve_fullspace(spec, d["u_d"][b], d["y_d"][b], up[b], yp[b])
        assert np.max(np.abs(res["auto"][0][b] - sol.optimal_u)) / max(np.max(np.abs(sol.optimal_u)), 1e-3) < TOL_U
        assert abs(res["auto"][1][b] - sol.cost) <= TOL_COST * max(abs(sol.cost), 1e-6)


def test_auto_refinement_threshold_extremes(gpu):
    # DDMPC_OPT_REFINE_RES_LOG10: threshold 1 (value 0) never refines -> equals OFF; threshold 1e-300 (value 3000): the bound
    # never decides, every instance is checked exactly and flagged -> equals ALWAYS; on the benchmark data the default flags
    # nothing (bound or exact check) -> equals OFF bit for bit
    spec = orc.spec_from_params()
    B = 24
    d = harness.generate_batch(range(300, 300 + B))
    up = d["u_d"][:, -4:, :].reshape(B, -1).copy(); yp = d["y_d"][:, -4:, :].reshape(B, -1).copy()
    out = {}
    with _spec_engine(spec, 400, B) as eng:
        eng.set_data(d["u_d"], d["y_d"])
        for tag, mode, thr in (("off", "off", None), ("never", "auto", 0.0), ("default", "auto", -10.7), ("all", "auto", -300.0),
                               ("always", "always", None)):
            eng.set_refinement(mode, res_log10=thr)
            out[tag] = tuple(x.copy() for x in eng.solve(up, yp))
        with pytest.raises(L.DDMPCError):
            eng.set_refinement("auto", res_log10=-301.0)
    for tag in ("never", "default"):
        assert np.array_equal(out[tag][0], out["off"][0]) and np.array_equal(out[tag][1], out["off"][1]), tag
    assert np.array_equal(out["all"][0], out["always"][0]) and np.array_equal(out["all"][1], out["always"][1])
    assert not np.array_equal(out["always"][0], out["off"][0])                   # refinement does change the last bits
    assert np.max(np.abs(out["always"][0] - out["off"][0])) < 1e-10 * np.max(np.abs(out["off"][0]))


# ------------------------------------------------------------------ warm steps beyond the register-resident kernels (NOMINAL)
def test_large_nominal_warm_step_reuses_the_factors(gpu):
    # cfg-5-like shape (m = p = 8, n = 8, L = 30: 608 rows, exact data): ddmpc_prepare forms everything that depends on the
    # data alone (Gram, its rank-revealing factor, C'WC and its factor) once; ddmpc_step then runs the substitutions and
    # the refinement passes on those factors -- the same arithmetic as a full solve, so the results must be BIT-equal to
    # ddmpc_solve's for every past window, before and after a change of the data set
    B = 6
    spec, plant, N, d, up, yp = _config5(B)
    # a consistent second past window: the last n steps of ANOTHER stretch of the same trajectory
    n = spec.n
    up2 = d["u_d"][:, 100:100 + n, :].reshape(B, -1).copy(); yp2 = d["y_d"][:, 100:100 + n, :].reshape(B, -1).copy()
    d2 = harness.generate_batch(range(50, 50 + B), N=N, plant=plant)
    with _spec_engine(spec, N, B) as eng:
        eng.set_data(d["u_d"], d["y_d"])
        cold1 = tuple(x.copy() for x in eng.solve(up, yp))
        cold2 = tuple(x.copy() for x in eng.solve(up2, yp2))
        eng.prepare()
        warm2 = tuple(x.copy() for x in eng.step(up2, yp2))
        ub_w = eng.get_solution("ubar"); al_w = eng.get_solution("alpha")
        warm1 = tuple(x.copy() for x in eng.step(up, yp))
        assert np.all(cold1[2] == 0) and np.all(cold2[2] == 0)
        for c, w in ((cold1, warm1), (cold2, warm2)):
            assert np.array_equal(c[0], w[0]) and np.array_equal(c[1], w[1]) and np.array_equal(c[2], w[2])
        assert not np.array_equal(cold1[0], cold2[0])
        assert np.array_equal(ub_w[:, n * spec.m:], warm2[0]) and np.all(np.isfinite(al_w))
        # new data: the factors are dropped with the data they belong to
        eng.set_data(d2["u_d"], d2["y_d"])
        upn = d2["u_d"][:, -n:, :].reshape(B, -1).copy(); ypn = d2["y_d"][:, -n:, :].reshape(B, -1).copy()
        warm_new = tuple(x.copy() for x in eng.step(upn, ypn))          # prepares on first use
        cold_new = tuple(x.copy() for x in eng.solve(upn, ypn))
        assert np.array_equal(warm_new[0], cold_new[0]) and np.array_equal(warm_new[1], cold_new[1])
        assert not np.array_equal(warm_new[0], warm1[0])
    # and the warm results are right: against the model-based solution
    from oracle.nominal_exact import solve_nominal_model_based
    for b in range(B):
        mod = solve_nominal_model_based(spec, plant, up2[b], yp2[b])
        assert np.max(np.abs(warm2[0][b] - mod["optimal_u"])) / np.max(np.abs(mod["optimal_u"])) < TOL_U
        assert abs(warm2[1][b] - mod["cost"]) <= TOL_COST * abs(mod["cost"])


def test_large_nominal_closed_loop_warm_equals_cold(gpu):
    # the per-step closed loop of a NOMINAL controller at that size: with the factors of ddmpc_prepare (default path) and with a
    # full solve per step (DDMPC_PATH_COLD) the trajectories must be bit-equal
    B, n_steps = 4, 24
    spec, plant, N, d, up, yp = _config5(B)
    w = np.zeros((B, n_steps, spec.p))
    out = {}
    for path in ("auto", "cold"):
        with _spec_engine(spec, N, B) as eng:
            eng.set_data(d["u_d"], d["y_d"])
            eng.set_closed_loop_path(path)
            out[path] = eng.closed_loop(plant["A"], plant["B"], plant["C"], plant["D"], d["x_end"], up, yp, w, n_mpc_step=1)
    for a, b in zip(out["auto"], out["cold"]):
        assert np.array_equal(a, b)
    u_sys, y_sys, status = out["auto"][:3]
    assert np.all(status == 0)
    assert np.max(np.abs(y_sys[:, -1, :] - spec.y_s)) < np.max(np.abs(y_sys[:, 0, :] - spec.y_s))      # it is heading for the setpoint


@pytest.mark.parametrize("slack", ["none", "convex"])
def test_large_robust_warm_step_reuses_the_factors(gpu, slack):
    # ROBUST scheme beyond the register-resident kernels (m = p = 3, L = 44: 288 rows): ddmpc_prepare keeps Gram + lam D, the
    # factor of the columns outside the slack box and the Schur complement of the boxed block; ddmpc_step runs the rest
    # (substitutions, active-set iterations on the boxed block, refinement) on them -- bit-equal to ddmpc_solve, for two
    # past windows and after a change of the data set; and right: against the full-space oracle
    rng = np.random.default_rng(11)
    m = p = 3; ns = n = 4; Lh = 44; N = 700; B = 4
    import test_gpu_parity as T
    plant = T._random_plant(rng, ns, m, p, 0.002)
    spec = orc.QPSpec(n=n, m=m, p=p, L=Lh, Q=2.0 * np.eye(p * Lh), R=0.05 * np.eye(m * Lh), u_s=0.1 * np.ones(m),
                      y_s=0.05 * np.ones(p), robust=True, eps_max=0.002, lamb_alpha=30.0, lamb_sigma=800.0, c=1.0, slack=slack, tec=True)
    d = harness.generate_batch(range(70, 70 + B), N=N, plant=plant)
    d2 = harness.generate_batch(range(90, 90 + B), N=N, plant=plant)
    up = d["u_d"][:, -n:, :].reshape(B, -1).copy(); yp = d["y_d"][:, -n:, :].reshape(B, -1).copy()
    up2 = d["u_d"][:, 50:50 + n, :].reshape(B, -1).copy(); yp2 = d["y_d"][:, 50:50 + n, :].reshape(B, -1).copy()
    with _spec_engine(spec, N, B) as eng:
        assert "large_solve" in eng.kernel_name()
        eng.set_data(d["u_d"], d["y_d"])
        cold1 = tuple(x.copy() for x in eng.solve(up, yp))
        cold2 = tuple(x.copy() for x in eng.solve(up2, yp2))
        eng.prepare()
        warm2 = tuple(x.copy() for x in eng.step(up2, yp2))
        sg = eng.get_solution("sigma")
        warm1 = tuple(x.copy() for x in eng.step(up, yp))
        for c, w in ((cold1, warm1), (cold2, warm2)):
            assert np.all(c[2] == 0)
            for k in range(4):
                assert np.array_equal(c[k], w[k]), k
        assert not np.array_equal(cold1[0], cold2[0]) and np.all(np.isfinite(sg))
        eng.set_data(d2["u_d"], d2["y_d"])
        upn = d2["u_d"][:, -n:, :].reshape(B, -1).copy(); ypn = d2["y_d"][:, -n:, :].reshape(B, -1).copy()
        warm_new = tuple(x.copy() for x in eng.step(upn, ypn))          # prepares on first use
        cold_new = tuple(x.copy() for x in eng.solve(upn, ypn))
        assert np.array_equal(warm_new[0], cold_new[0]) and np.array_equal(warm_new[1], cold_new[1])
    for b in range(B):
        sol = orc.solve_fullspace(spec, d["u_d"][b], d["y_d"][b], up2[b], yp2[b])
        assert sol.status == "optimal" and (slack == "none" or int(warm2[3][b]) == sol.iters)
        assert np.max(np.abs(warm2[0][b] - sol.optimal_u)) / np.max(np.abs(sol.optimal_u)) < TOL_U
        assert abs(warm2[1][b] - sol.cost) <= TOL_COST * abs(sol.cost)


def test_large_robust_closed_loop_warm_equals_cold(gpu):
    # per-step closed loop of a ROBUST controller beyond the register-resident kernels (slack box): the default path (solves on
    # what ddmpc_prepare kept) and a full solve per step (DDMPC_PATH_COLD) give bit-equal trajectories
    rng = np.random.default_rng(11)
    m = p = 3; ns = n = 4; Lh = 44; N = 700; B = 3; n_steps = 12
    import test_gpu_parity as T
    plant = T._random_plant(rng, ns, m, p, 0.002)
    spec = orc.QPSpec(n=n, m=m, p=p, L=Lh, Q=2.0 * np.eye(p * Lh), R=0.05 * np.eye(m * Lh), u_s=0.1 * np.ones(m),
                      y_s=0.05 * np.ones(p), robust=True, eps_max=0.002, lamb_alpha=30.0, lamb_sigma=800.0, c=1.0, slack="convex", tec=True)
    d = harness.generate_batch(range(70, 70 + B), N=N, plant=plant)
    up = d["u_d"][:, -n:, :].reshape(B, -1).copy(); yp = d["y_d"][:, -n:, :].reshape(B, -1).copy()
    w = 0.002 * np.random.default_rng(5).uniform(-1, 1, size=(B, n_steps, p))
    out = {}
    for path in ("auto", "cold"):
        with _spec_engine(spec, N, B) as eng:
            eng.set_data(d["u_d"], d["y_d"])
            eng.set_closed_loop_path(path)
            out[path] = eng.closed_loop(plant["A"], plant["B"], plant["C"], plant["D"], d["x_end"], up, yp, w, n_mpc_step=1)
    for a, b in zip(out["auto"], out["cold"]):
        assert np.array_equal(a, b)
    assert np.all(out["auto"][2] == 0)
