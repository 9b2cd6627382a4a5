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


# ------------------------------------------------------------------ cfg 5 at its stated batch, phase-kernel pipeline
def test_config5_at_its_stated_batch(gpu):
    # BASELINE configs[4] (nominal, m = p = 8, n = 8, L = 30, N = 2000, exact data; 608 rows, rank 312), ALL 512 instances, on
    # the default path (phase kernels: ddmpc_rr2.hpp / ddmpc_rr2_solve.hpp), every one against the model-based solution of
    # the same QP at the standard bars; ubar of the solution consistent with optimal_u
    from test_gpu_round3 import _config5
    from oracle.nominal_exact import solve_nominal_model_based_batch
    B = 512
    spec, plant, N, d, up, yp = _config5(B)
    u_ref, c_ref, feas = solve_nominal_model_based_batch(spec, plant, up, yp)
    assert np.max(feas) < 1e-10
    with _spec_engine(spec, N, B) as eng:
        eng.set_data(d["u_d"], d["y_d"])
        u, cost, status, iters = eng.solve(up, yp)
        ub = eng.get_solution("ubar")
    assert np.all(status == 0) and np.all(iters == 1)
    eu = np.max(np.max(np.abs(u - u_ref), axis=1) / np.max(np.abs(u_ref), axis=1))
    ec = np.max(np.abs(cost - c_ref) / np.abs(c_ref))
    assert eu < TOL_U and ec < TOL_COST, (eu, ec)
    assert np.array_equal(ub[:, spec.n * spec.m:], u)


def _exact_plant_case(seed, m, p, n, Lh, N, B):
    rng = np.random.default_rng(seed)
    ns = n
    A = rng.normal(size=(ns, ns)); A *= 0.9 / max(abs(np.linalg.eigvals(A)))
    plant = dict(A=A, B=rng.normal(size=(ns, m)), C=rng.normal(size=(p, ns)), D=np.zeros((p, m)), eps_max=0.0)
    u_s = 0.1 * np.ones(m); y_s = (plant["C"] @ np.linalg.inv(np.eye(ns) - A) @ plant["B"]) @ u_s
    spec = orc.QPSpec(n=n, m=m, p=p, L=Lh, Q=3.0 * np.eye(p * Lh), R=1e-4 * np.eye(m * Lh), u_s=u_s, y_s=y_s, robust=False,
                      eps_max=0.0, lamb_alpha=0.0, lamb_sigma=0.0, c=0.0, slack="none", tec=True)
    d = harness.generate_batch(range(B), N=N, plant=plant)
    up = d["u_d"][:, -n:, :].reshape(B, -1).copy(); yp = d["y_d"][:, -n:, :].reshape(B, -1).copy()
    return spec, plant, d, up, yp


@pytest.mark.parametrize("shape", [(2, 3, 3, 60, 900), (2, 2, 4, 70, 700), (5, 4, 5, 40, 1200)],
                         ids=["5ch-315rows", "4ch-296rows", "9ch-405rows"])
def test_large_nominal_pipelines_agree(gpu, shape):
    # NOMINAL controllers beyond the register-resident kernels on exact data of random stable plants -- channel counts that
    # do not fill a 16-wide tile, row counts that are no multiple of 16 or 64, fixed blocks that end inside a panel: the phase
    # kernels (default) and the one-workgroup kernels (DDMPC_OPT_LARGE_PIPELINE) must agree with each other far inside the
    # bars and both meet the model-based solution of the same QP (controller.py:506-538,679-711)
    from oracle.nominal_exact import solve_nominal_model_based_batch
    m, p, n, Lh, N = shape
    B = 5
    # (plant seeds 4 / 125 / 142: dependent rows leave <= 1e-11 of the largest Gram diagonal as rounding residue, far below the
    #  rank tolerance 1e-8; DESIGN.md section 9 records a 9-channel plant whose residue sits AT the tolerance)
    spec, plant, d, up, yp = _exact_plant_case({9: 4, 5: 125, 4: 142}[m + p], m, p, n, Lh, N, B)
    u_ref, c_ref, feas = solve_nominal_model_based_batch(spec, plant, up, yp)
    assert np.max(feas) < 1e-10
    res = {}
    for mode in ("phases", "one_workgroup"):
        with _spec_engine(spec, N, B) as eng:
            assert (m + p) * (Lh + n) > 271 and "nominal_rr" in eng.kernel_name()
            eng.set_large_pipeline(mode)
            eng.set_data(d["u_d"], d["y_d"])
            res[mode] = tuple(x.copy() for x in eng.solve(up, yp))
            al = eng.get_solution("alpha"); ub = eng.get_solution("ubar"); yb = eng.get_solution("ybar")
            # alpha (controller.py:434): H alpha reproduces [ubar; ybar]
            Hu, Hy = orc.hankel_matrix(d["u_d"][0], spec.Ln), orc.hankel_matrix(d["y_d"][0], spec.Ln)
            sc = max(np.max(np.abs(ub[0])), np.max(np.abs(yb[0])))
            assert np.max(np.abs(Hu @ al[0] - ub[0])) < 1e-7 * sc and np.max(np.abs(Hy @ al[0] - yb[0])) < 1e-7 * sc, mode
    for mode, (u, c, st, it) in res.items():
        assert np.all(st == 0), mode
        eu = np.max(np.max(np.abs(u - u_ref), axis=1) / np.max(np.abs(u_ref), axis=1))
        ec = np.max(np.abs(c - c_ref) / np.abs(c_ref))
        assert eu < TOL_U and ec < TOL_COST, (mode, eu, ec)
    a, bq = res["phases"], res["one_workgroup"]
    assert np.max(np.abs(a[0] - bq[0])) < 1e-8 * np.max(np.abs(bq[0])) and np.max(np.abs(a[1] - bq[1]) / np.abs(bq[1])) < 1e-10


def test_large_nominal_infeasible_setpoint_and_refinement_cap(gpu):
    # exact data and a terminal equality the plant cannot meet (a setpoint that is no equilibrium): CVXPY would report
    # "infeasible" (controller.py:585-629), and so must both pipelines; with a true equilibrium the result does not depend on
    # the cap on refinement passes (no instance of this data asks for a second one)
    m, p, n, Lh, N, B = 3, 3, 4, 44, 900, 4
    spec, plant, d, up, yp = _exact_plant_case(7, m, p, n, Lh, N, B)
    bad = orc.QPSpec(**{**spec.__dict__, "y_s": spec.y_s + 0.3})
    for mode in ("phases", "one_workgroup"):
        with _spec_engine(bad, N, B) as eng:
            eng.set_large_pipeline(mode)
            eng.set_data(d["u_d"], d["y_d"])
            _, _, st, _ = eng.solve(up, yp)
            assert [L.STATUS_STRINGS[int(s)] for s in st] == ["infeasible"] * B, mode
    out = {}
    for cap in (1, 3):
        with _spec_engine(spec, N, B) as eng:
            eng.set_refinement("auto", max_passes=cap)
            eng.set_data(d["u_d"], d["y_d"])
            out[cap] = tuple(x.copy() for x in eng.solve(up, yp))
    assert np.all(out[1][2] == 0) and np.array_equal(out[1][0], out[3][0]) and np.array_equal(out[1][1], out[3][1])


# ------------------------------------------------------------------ the affine law beyond the register-resident kernels
def test_large_nominal_affine_law(gpu):
    # DDMPC_OPT_LARGE_AFFINE_LAW on the cfg-5 shape: ddmpc_prepare forms z(past) from solves at the zero window and the n(m+p)
    # unit windows; ddmpc_step then evaluates it -- within the bars of the cold solve (controller.py:389-407 with the data fixed)
    # for consistent past windows, "infeasible" for a window no trajectory of the plant explains; ddmpc_get_gain returns the
    # law itself: z = [ubar; ybar] per component; the per-step closed loop on the law follows the one on the kept factors
    from test_gpu_round3 import _config5
    B = 6
    spec, plant, N, d, up, yp = _config5(B)
    n, m, p = spec.n, spec.m, spec.p
    up2 = d["u_d"][:, 100:100 + n, :].reshape(B, -1).copy(); yp2 = d["y_d"][:, 100:100 + n, :].reshape(B, -1).copy()
    bad_y = yp2 + 0.05                                                   # no trajectory of the plant has this past window
    with _spec_engine(spec, N, B) as eng:
        with pytest.raises(L.DDMPCError):
            eng.gain()                                                   # no law without the option
        eng.set_large_affine_law(True)
        eng.set_data(d["u_d"], d["y_d"])
        cold = tuple(x.copy() for x in eng.solve(up2, yp2))
        cold_bad = tuple(x.copy() for x in eng.solve(up2, bad_y))
        eng.prepare()
        warm = tuple(x.copy() for x in eng.step(up2, yp2))
        ub, yb = eng.get_solution("ubar"), eng.get_solution("ybar")
        al = eng.get_solution("alpha")                                   # (a full solve on the factors behind the scenes)
        warm_bad = tuple(x.copy() for x in eng.step(up2, bad_y))
        warm1 = tuple(x.copy() for x in eng.step(up, yp))
        g = eng.gain()
    assert np.all(cold[2] == 0) and np.all(warm[2] == 0)
    sc = np.max(np.abs(cold[0]), axis=1)
    assert np.max(np.max(np.abs(warm[0] - cold[0]), axis=1) / sc) < TOL_U
    assert np.max(np.abs(warm[1] - cold[1]) / np.abs(cold[1])) < TOL_COST
    assert [L.STATUS_STRINGS[int(s)] for s in cold_bad[2]] == ["infeasible"] * B == [L.STATUS_STRINGS[int(s)] for s in warm_bad[2]]
    assert np.all(warm1[2] == 0) and not np.array_equal(warm1[0], warm[0])
    # the law itself: component order (time-major, the m + p channels of a step adjacent)
    nf, r = n * (m + p), (m + p) * (spec.L + n)
    assert g.shape == (B, nf + 1, r)
    past = np.concatenate([up2, yp2], axis=1)
    z = g[:, 0, :] + np.einsum("bjr,bj->br", g[:, 1:, :], past)
    zz = z.reshape(B, spec.L + n, m + p)
    assert np.max(np.abs(zz[:, :, :m].reshape(B, -1) - ub)) < 1e-9 * np.max(np.abs(ub))
    assert np.max(np.abs(zz[:, :, m:].reshape(B, -1) - yb)) < 1e-9 * np.max(np.abs(yb))
    assert np.array_equal(ub[:, n * m:], warm[0]) and np.all(np.isfinite(al))
    # closed loop: on the law vs on the kept factors
    n_steps = 16
    w = np.zeros((B, n_steps, p))
    out = {}
    for law in (False, True):
        with _spec_engine(spec, N, B) as eng:
            eng.set_large_affine_law(law)
            eng.set_data(d["u_d"], d["y_d"])
            out[law] = eng.closed_loop(plant["A"], plant["B"], plant["C"], plant["D"], d["x_end"], up, yp, w, n_mpc_step=1)
    assert np.all(out[True][2] == 0)
    assert np.max(np.abs(out[True][0] - out[False][0])) < 1e-7 * np.max(np.abs(out[False][0]))
    assert np.max(np.abs(out[True][1] - out[False][1])) < 1e-7 * np.max(np.abs(out[False][1]))


# ------------------------------------------------------------------ cfg 5 against a reference-formulation oracle beyond fp64
def test_config5_against_the_extended_precision_golden_solutions(gpu):
    # tests/golden/cfg5_extended.npz: sixteen instances of BASELINE configs[4] -- among them 283, the one on which the fp64 SVD
    # route and the model-based solution disagree at the 1e-8 level -- solved from the DATA alone (the reference's formulation,
    # controller.py:506-538,549-629,679-711; no (A, B, C)) in 80-bit arithmetic with orthogonal factorisations
    # (tests/golden/make_golden_cfg5.py, good to ~1e-13).  The GPU (default path: phase kernels) against THAT at the standard bars.
    from test_gpu_round3 import _config5
    z = np.load(os.path.join(ROOT, "tests", "golden", "cfg5_extended.npz"))
    inst = [int(b) for b in z["instances"]]
    B = 512
    spec, plant, N, d, up, yp = _config5(B)
    with _spec_engine(spec, N, B) as eng:
        eng.set_data(d["u_d"], d["y_d"])
        u, cost, status, _ = eng.solve(up, yp)
        res = {}
        for mode in ("one_workgroup",):
            eng.set_large_pipeline(mode)
            res[mode] = tuple(x.copy() for x in eng.solve(up, yp))
    assert np.all(status == 0)
    eu = np.array([np.max(np.abs(u[b] - z["optimal_u"][k])) / np.max(np.abs(z["optimal_u"][k])) for k, b in enumerate(inst)])
    ec = np.array([abs(cost[b] - z["cost"][k]) / z["cost"][k] for k, b in enumerate(inst)])
    print("cfg 5 vs extended-precision golden: max rel err u %.2e (instance 283: %.2e), cost %.2e" % (eu.max(), eu[0], ec.max()))
    assert inst[0] == 283 and eu.max() < TOL_U and ec.max() < TOL_COST, (eu, ec)
    uo = res["one_workgroup"][0]
    eo = max(np.max(np.abs(uo[b] - z["optimal_u"][k])) / np.max(np.abs(z["optimal_u"][k])) for k, b in enumerate(inst))
    assert eo < TOL_U, eo


@pytest.mark.gpu
def test_results_do_not_depend_on_what_the_allocator_hands_back():
    """Every work buffer is written before it is read: the same solves with fresh device buffers pre-filled with two different
    byte patterns (ddmpc_debug_poison_allocations) and not pre-filled -- then they hold whatever earlier tests of this process
    left in the memory the allocator hands back -- give bit-identical outputs.  (A short trajectory used to leave three of the
    eight partial-sum slots of H(H'x) unwritten and summed: correct on a fresh box, where device memory comes back zeroed, off
    by up to 1e-8 after other controllers had used the memory.)"""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _poison_probe
    digests = [_poison_probe.digest(f) for f in (63, 255, 0, 64)]
    assert len(set(digests)) == 1, digests


@pytest.mark.gpu
@pytest.mark.parametrize("m,p", [(1, 1), (2, 1), (1, 2), (3, 2), (2, 3), (3, 3), (4, 4), (5, 3)])
def test_structured_gram_for_any_channel_count(m, p):
    """hankel_matrix.py:5-53 is generic in the channel count, and so is the Hankel-structured Gram now: for m + p != 4 the
    register-resident kernels take G = H H' from a launch ahead of them (ddmpc_gram_tiles_kernel: first rows by MFMA, the rest by
    the sliding-window recurrence; round 5: rr2_gram_tiles*_kernel) instead of the dense r^2 c product.  Both Gram modes against the full-space oracle and against each other,
    ROBUST with the slack box (active-set iterations reload the tiles) and without, through ddmpc_solve, the chunked
    ddmpc_solve_from_host and ddmpc_prepare / ddmpc_step."""
    from direct_data_driven_mpc_amd.harness import generate_batch
    rng = np.random.default_rng(40 + 7 * m + p)
    ns = n = 3
    nch = m + p
    Lh = max(2 * n, 132 // nch - n)                       # ~130 rows: the 9-tile instance of the kernels (or the one below it)
    N = (m + 1) * (Lh + 2 * n) + 150
    A = rng.normal(size=(ns, ns)); A *= 0.85 / max(abs(np.linalg.eigvals(A)))
    plant = dict(A=A, B=rng.normal(size=(ns, m)), C=rng.normal(size=(p, ns)), D=np.zeros((p, m)), eps_max=0.002)
    B = 5
    d = generate_batch(range(B), N=N, plant=plant)
    up = d["u_d"][:, -n:, :].reshape(B, -1).copy(); yp = d["y_d"][:, -n:, :].reshape(B, -1).copy()
    for slack in ("none", "convex"):
        spec = orc.QPSpec(n=n, m=m, p=p, L=Lh, Q=2.0 * np.eye(p * Lh), R=0.05 * np.eye(m * Lh), u_s=rng.uniform(-0.5, 0.5, m),
                          y_s=rng.uniform(-0.5, 0.5, p), robust=True, eps_max=0.002, lamb_alpha=20.0, lamb_sigma=500.0, c=1.0,
                          slack=slack, tec=True)
        res = {}
        for mode in (L.GRAM_DENSE, L.GRAM_STRUCTURED, L.GRAM_AUTO):
            with BatchedDDMPC(n=n, m=m, p=p, L_=Lh, N=N, Q=2.0, R=0.05, u_s=spec.u_s, y_s=spec.y_s, batch=B, controller_type=L.ROBUST,
                              slack_type=L.SLACK_CONVEX if slack == "convex" else L.SLACK_NONE, eps_max=0.002, lamb_alpha=20.0,
                              lamb_sigma=500.0, c=1.0, gram_mode=mode) as eng:
                assert "cold" in eng.kernel_name()
                eng.set_refinement("always")
                eng.set_data(d["u_d"], d["y_d"])
                u, cost, status, iters = eng.solve(up, yp)
                uh = eng.solve_from_host(d["u_d"], d["y_d"], up, yp)
                eng.set_data(d["u_d"], d["y_d"])
                uw = eng.step(up, yp)
                flops, _ = eng.cost_model()
            res[mode] = (u, cost, status, iters, flops)
            assert np.array_equal(uh[0], u) and np.array_equal(uh[2], status)
            assert np.max(np.abs(uw[0] - u)) <= 1e-8 * np.max(np.abs(u)) and np.array_equal(uw[2], status)
            for b in range(B):
                sol = orc.solve_fullspace(spec, d["u_d"][b], d["y_d"][b], up[b], yp[b])
                assert L.STATUS_STRINGS[int(status[b])] == sol.status == "optimal"
                assert np.max(np.abs(u[b] - sol.optimal_u)) / max(np.max(np.abs(sol.optimal_u)), 1e-3) < 1e-8, (mode, b)
                assert abs(cost[b] - sol.cost) <= 1e-8 * max(abs(sol.cost), 1e-6), (mode, b)
        dn, st = res[L.GRAM_DENSE], res[L.GRAM_STRUCTURED]
        assert np.array_equal(dn[3], st[3])                                    # the same active-set iterations in both modes
        assert np.max(np.abs(dn[0] - st[0])) <= 1e-9 * np.max(np.abs(dn[0]))
        assert np.array_equal(res[L.GRAM_AUTO][0], st[0])                      # AUTO is the structured Gram for every channel count
        assert dn[4] > 1.5 * st[4]                                             # the dense Gram is charged r^2 c flops
        if nch not in (2, 4):
            # both launches that can form the tiles ahead of the kernel (DDMPC_OPT_GRAM_LAUNCH; round 5: the streaming
            # matrix-pipe launch with several lags per tile, the default from six channels on)
            for launch in ("matrix_pipe", "staged"):
                with BatchedDDMPC(n=n, m=m, p=p, L_=Lh, N=N, Q=2.0, R=0.05, u_s=spec.u_s, y_s=spec.y_s, batch=B, controller_type=L.ROBUST,
                                  slack_type=L.SLACK_CONVEX if slack == "convex" else L.SLACK_NONE, eps_max=0.002, lamb_alpha=20.0,
                                  lamb_sigma=500.0, c=1.0, gram_mode=L.GRAM_STRUCTURED) as eng:
                    eng.set_refinement("always")
                    eng.set_gram_launch(launch)
                    eng.set_data(d["u_d"], d["y_d"])
                    u2, cost2, status2, iters2 = eng.solve(up, yp)
                assert np.array_equal(iters2, st[3]) and np.array_equal(status2, st[2]), launch
                assert np.max(np.abs(u2 - st[0])) <= 1e-9 * np.max(np.abs(st[0])), launch
                assert np.max(np.abs(cost2 - st[1]) / np.abs(st[1])) <= 1e-9, launch


@pytest.mark.gpu
@pytest.mark.parametrize("slack", ["none", "convex"])
def test_dense_weights_with_unweighted_components(slack):
    """controller.py:708-710 takes any PSD `Q`, `R`.  Dense matrices whose null space is spanned by coordinate axes -- some
    components carry no weight at all: all-zero rows and columns -- are served like zeros on the diagonal of a DIAG matrix:
    those components leave the block that is inverted and get 1/w = 1e25 (ddmpc_api.hip upload_params).  Against the
    full-space oracle, which takes the singular matrices as they are.  (A null space off the axes stays DDMPC_ERR_UNSUPPORTED:
    second half.)"""
    from direct_data_driven_mpc_amd.harness import generate_batch
    rng = np.random.default_rng(77)
    spec = orc.spec_from_params(slack_var_constraint_type=1 if slack == "convex" else 0)
    nq, nr = spec.p * spec.L, spec.m * spec.L

    def spd(n, scale, k):
        A = rng.normal(size=(n, n)) / np.sqrt(n)
        return scale * (np.eye(n) + k * 0.1 * (A + A.T) / 2.0)
    Q = spd(nq, 3.0, 3); R = spd(nr, 1e-2, 2)
    for i in (5, 6, 17, 40, nq - 1):                       # outputs without a weight (the last one sits in the terminal steps)
        Q[i, :] = 0.0; Q[:, i] = 0.0
    for i in (9, 30):                                      # inputs without a weight
        R[i, :] = 0.0; R[:, i] = 0.0
    assert np.linalg.matrix_rank(Q) == nq - 5
    spec.Q, spec.R = Q, R
    B = 4
    d = generate_batch(range(B), N=400)
    up = d["u_d"][:, -4:, :].reshape(B, -1).copy(); yp = d["y_d"][:, -4:, :].reshape(B, -1).copy()
    with BatchedDDMPC(n=spec.n, m=spec.m, p=spec.p, L_=spec.L, N=400, Q=Q, R=R, u_s=spec.u_s, y_s=spec.y_s, batch=B,
                      controller_type=L.ROBUST, slack_type=L.SLACK_CONVEX if slack == "convex" else L.SLACK_NONE,
                      eps_max=spec.eps_max, lamb_alpha=spec.lamb_alpha, lamb_sigma=spec.lamb_sigma, c=spec.c) as eng:
        eng.set_data(d["u_d"], d["y_d"])
        u, cost, status, iters = eng.solve(up, yp)
        uw = eng.step(up, yp)
    for b in range(B):
        sol = orc.solve_fullspace(spec, d["u_d"][b], d["y_d"][b], up[b], yp[b])
        assert L.STATUS_STRINGS[int(status[b])] == sol.status == "optimal"
        assert np.max(np.abs(u[b] - sol.optimal_u)) / np.max(np.abs(sol.optimal_u)) < 1e-8, b
        assert abs(cost[b] - sol.cost) <= 1e-9 * abs(sol.cost), b
        if slack == "convex":
            assert int(iters[b]) == sol.iters
    assert np.max(np.abs(uw[0] - u)) <= 1e-8 * np.max(np.abs(u))
    # a null direction that is no coordinate axis: still refused, with a message that says why
    v = np.zeros(nq); v[[3, 4]] = (1.0, -1.0)
    Qs = spd(nq, 3.0, 3); Qs = Qs - np.outer(Qs @ v, Qs @ v) / (v @ Qs @ v)          # Qs v = 0, PSD
    with pytest.raises(L.DDMPCError, match="unweighted components"):
        BatchedDDMPC(n=spec.n, m=spec.m, p=spec.p, L_=spec.L, N=400, Q=Qs, R=R, u_s=spec.u_s, y_s=spec.y_s, batch=1,
                     controller_type=L.ROBUST, slack_type=L.SLACK_NONE, eps_max=spec.eps_max, lamb_alpha=spec.lamb_alpha,
                     lamb_sigma=spec.lamb_sigma, c=spec.c)


@pytest.mark.gpu
@pytest.mark.parametrize("rows", [24, 44, 76, 108, 140, 204, 260])
def test_siso_structured_gram_on_every_kernel_instance(rows):
    """Two channels ride on the four-channel structured Gram inside the cold-solve kernel (a SISO Hankel matrix = two interleaved
    four-channel ones over the same flat trajectory).  Every instance of the kernel (<2,1> ... <17,8>), odd and even
    L + n (an odd one leaves half a pseudo time step of padding rows), odd and even column counts (the two interleaved
    trajectories then differ in length): structured against the dense product and both against the full-space oracle."""
    from direct_data_driven_mpc_amd.harness import generate_batch
    rng = np.random.default_rng(rows)
    m = p = 1; n = 3
    Ln = rows // 2                                            # 12, 22, 38, 54, 70, 102, 130 (even) ...
    if rows % 8 == 4: Ln += 1                                 # ... and a few odd ones: 23, 55, 71, 131
    Lh = Ln - n
    N = 2 * (Lh + 2 * n) + 160 + (rows // 4) % 2              # odd and even numbers of Hankel columns
    A = rng.normal(size=(n, n)); A *= 0.8 / max(abs(np.linalg.eigvals(A)))
    plant = dict(A=A, B=rng.normal(size=(n, m)), C=rng.normal(size=(p, n)), D=np.zeros((p, m)), eps_max=0.002)
    spec = orc.QPSpec(n=n, m=m, p=p, L=Lh, Q=2.0 * np.eye(p * Lh), R=0.05 * np.eye(m * Lh), u_s=np.array([0.3]), y_s=np.array([-0.2]),
                      robust=True, eps_max=0.002, lamb_alpha=20.0, lamb_sigma=500.0, c=1.0, slack="convex" if rows % 16 == 12 else "none", tec=True)
    B = 3
    d = generate_batch(range(B), N=N, plant=plant)
    up = d["u_d"][:, -n:, :].reshape(B, -1).copy(); yp = d["y_d"][:, -n:, :].reshape(B, -1).copy()
    res = {}
    for mode in (L.GRAM_DENSE, L.GRAM_STRUCTURED):
        with BatchedDDMPC(n=n, m=m, p=p, L_=Lh, N=N, Q=2.0, R=0.05, u_s=spec.u_s, y_s=spec.y_s, batch=B, controller_type=L.ROBUST,
                          slack_type=L.SLACK_CONVEX if spec.slack == "convex" else L.SLACK_NONE, eps_max=0.002, lamb_alpha=20.0,
                          lamb_sigma=500.0, c=1.0, gram_mode=mode) as eng:
            eng.set_refinement("always")
            eng.set_data(d["u_d"], d["y_d"])
            res[mode] = tuple(x.copy() for x in eng.solve(up, yp)) + (eng.kernel_name(), eng.cost_model()[0])
    dn, st = res[L.GRAM_DENSE], res[L.GRAM_STRUCTURED]
    assert dn[4] == st[4] and "cold" in st[4]
    assert st[5] < dn[5]                                      # the structured mode is charged the structured flop count
    for b in range(B):
        sol = orc.solve_fullspace(spec, d["u_d"][b], d["y_d"][b], up[b], yp[b])
        for r_ in (dn, st):
            assert L.STATUS_STRINGS[int(r_[2][b])] == sol.status == "optimal"
            assert np.max(np.abs(r_[0][b] - sol.optimal_u)) / max(np.max(np.abs(sol.optimal_u)), 1e-3) < 1e-8, (rows, b)
            assert abs(r_[1][b] - sol.cost) <= 1e-8 * max(abs(sol.cost), 1e-6), (rows, b)
    assert np.max(np.abs(dn[0] - st[0])) <= 1e-9 * np.max(np.abs(dn[0])) and np.array_equal(dn[3], st[3])


@pytest.mark.gpu
@pytest.mark.parametrize("case", [19, 21, 23, 35, 41, 47, 69, 83])
def test_large_nominal_seeded_plants_on_the_phase_kernels(case):
    """Eight plants of tools/nominal_fuzz.py beyond 271 rows (SISO with L up to 313, 1x3, 2x3; trajectories short enough that the
    matrix-pipe H(H'x) runs with fewer column groups than it has partial-sum slots -- case 35 is the one that exposed the
    unwritten slots; dependent row tiles retire early on all of them) on the phase kernels, against the model-based solution."""
    from direct_data_driven_mpc_amd.harness import generate_batch
    from oracle.nominal_exact import solve_nominal_model_based
    rng = np.random.default_rng(9000 + case)
    m, p = [(2, 2), (1, 3), (3, 1), (2, 3), (4, 2), (1, 1)][case % 6]
    ns = n = int(rng.integers(2, 5))
    rows = int(rng.integers(60, 260)) if case % 2 == 0 else int(rng.integers(280, 640))
    Lh = max(2 * n, rows // (m + p) - n)
    N = (m + 1) * (Lh + 2 * n) + int(rng.integers(100, 300))
    assert (m + p) * (Lh + n) > 271
    A = rng.normal(size=(ns, ns)); A *= rng.uniform(0.5, 0.9) / max(abs(np.linalg.eigvals(A)))
    plant = dict(A=A, B=rng.normal(size=(ns, m)), C=rng.normal(size=(p, ns)), D=np.zeros((p, m)), eps_max=0.0)
    u_s = rng.uniform(-0.5, 0.5, m)
    y_s = (plant["C"] @ np.linalg.inv(np.eye(ns) - A) @ plant["B"]) @ u_s
    q, rw = float(rng.uniform(1.0, 4.0)), float(rng.uniform(0.01, 0.2))
    spec = orc.QPSpec(n=n, m=m, p=p, L=Lh, Q=q * np.eye(p * Lh), R=rw * np.eye(m * Lh), u_s=u_s, y_s=y_s, robust=False,
                      eps_max=0.0, lamb_alpha=0.0, lamb_sigma=0.0, c=0.0, slack="none", tec=True)
    B = 2
    d = generate_batch(range(case * 10, case * 10 + B), N=N, plant=plant)
    up = d["u_d"][:, -n:, :].reshape(B, -1).copy(); yp = d["y_d"][:, -n:, :].reshape(B, -1).copy()
    with BatchedDDMPC(n=n, m=m, p=p, L_=Lh, N=N, Q=q, R=rw, u_s=u_s, y_s=y_s, batch=B, controller_type=L.NOMINAL) as eng:
        eng.set_large_pipeline("phases")
        eng.set_data(d["u_d"], d["y_d"])
        u, cost, status, _ = eng.solve(up, yp)
        uw = eng.step(up, yp)
        eng.set_large_affine_law(True)                        # ... and as an affine law of the past window (one launch per step)
        eng.prepare()
        ul = eng.step(up, yp)
        g = eng.gain()
    assert np.all(status == 0) and np.array_equal(uw[0], u) and np.array_equal(uw[1], cost)      # warm on the kept factors: bit-equal
    assert np.all(ul[2] == 0) and np.max(np.abs(ul[0] - u)) <= 1e-8 * np.max(np.abs(u)) and np.max(np.abs(ul[1] - cost)) <= 1e-9 * np.max(np.abs(cost)) + 1e-18
    assert g.shape == (B, n * (m + p) + 1, (m + p) * (Lh + n)) and np.all(np.isfinite(g))
    for b in range(B):
        mod = solve_nominal_model_based(spec, plant, up[b], yp[b])
        assert np.max(np.abs(u[b] - mod["optimal_u"])) / max(np.max(np.abs(mod["optimal_u"])), 1e-3) < 1e-8, (case, b)
        assert abs(cost[b] - mod["cost"]) <= 1e-9 * max(abs(mod["cost"]), 1e-9), (case, b)


@pytest.mark.gpu
@pytest.mark.parametrize("B", [1, 7])
def test_large_nominal_small_batches_and_switching_pipelines(B):
    """Batches of one and seven instances (grids of the lock-step launches with a single instance row; no multiple of anything),
    and DDMPC_OPT_LARGE_PIPELINE switched back and forth on a live handle between solves and around ddmpc_prepare: every solve
    agrees with the other pipeline's to rounding, a warm step is bit-equal to the cold solve of the pipeline that prepared."""
    from test_gpu_round3 import _config5
    spec, plant, N, d, up, yp = _config5(B)
    n = spec.n
    up2 = d["u_d"][:, 200:200 + n, :].reshape(B, -1).copy(); yp2 = d["y_d"][:, 200:200 + n, :].reshape(B, -1).copy()
    with _spec_engine(spec, N, B) as eng:
        eng.set_data(d["u_d"], d["y_d"])
        a = tuple(x.copy() for x in eng.solve(up, yp))                     # phases (default)
        eng.set_large_pipeline("one_workgroup")
        b = tuple(x.copy() for x in eng.solve(up, yp))
        eng.set_large_pipeline("phases")
        c = tuple(x.copy() for x in eng.solve(up, yp))
        eng.prepare()
        w = tuple(x.copy() for x in eng.step(up2, yp2))
        cold2 = tuple(x.copy() for x in eng.solve(up2, yp2))
        eng.set_large_pipeline("one_workgroup")                             # forgets what the other pipeline kept
        w1 = tuple(x.copy() for x in eng.step(up2, yp2))
        cold1 = tuple(x.copy() for x in eng.solve(up2, yp2))
    assert np.all(a[2] == 0) and np.all(b[2] == 0) and np.all(w[2] == 0) and np.all(w1[2] == 0)
    assert np.array_equal(a[0], c[0]) and np.array_equal(a[1], c[1])                       # the same pipeline: bit-equal
    sc = np.max(np.abs(a[0]), axis=1, keepdims=True)
    assert np.max(np.abs(a[0] - b[0]) / sc) < 1e-8 and np.max(np.abs(a[1] - b[1]) / np.abs(a[1])) < 1e-9
    assert np.array_equal(w[0], cold2[0]) and np.array_equal(w[1], cold2[1])
    assert np.array_equal(w1[0], cold1[0]) and np.array_equal(w1[1], cold1[1])
    assert np.max(np.abs(cold1[0] - cold2[0]) / np.max(np.abs(cold2[0]), axis=1, keepdims=True)) < 1e-8


@pytest.mark.gpu
@pytest.mark.parametrize("m,p,rows", [(2, 1, 30), (2, 1, 75), (1, 2, 111), (2, 1, 204), (1, 2, 261), (3, 2, 45), (3, 2, 105), (2, 3, 200), (3, 2, 265),
                                      (3, 3, 66), (3, 3, 258), (4, 3, 70), (4, 3, 266)])
def test_structured_gram_launch_on_every_kernel_instance(m, p, rows):
    """ddmpc_gram_tiles_kernel (the structured Gram of channel counts other than 2 and 4, formed ahead of the cold-solve kernel)
    on every instance of that kernel, <2,1> ... <17,8>: 3, 5, 6 and 7 channels, row counts off the tile size, structured against
    the dense product and both against the full-space oracle; warm step after ddmpc_prepare (the tiles are formed once there)."""
    from direct_data_driven_mpc_amd.harness import generate_batch
    rng = np.random.default_rng(100 * rows + m)
    n = 2
    nch = m + p
    Ln = rows // nch
    Lh = Ln - n
    assert Lh >= 2 * n
    N = (m + 1) * (Lh + 2 * n) + 120 + rows % 3
    A = rng.normal(size=(n, n)); A *= 0.8 / max(abs(np.linalg.eigvals(A)))
    plant = dict(A=A, B=rng.normal(size=(n, m)), C=rng.normal(size=(p, n)), D=np.zeros((p, m)), eps_max=0.002)
    slack = "convex" if rows % 2 else "none"
    spec = orc.QPSpec(n=n, m=m, p=p, L=Lh, Q=2.0 * np.eye(p * Lh), R=0.05 * np.eye(m * Lh), u_s=rng.uniform(-0.3, 0.3, m),
                      y_s=rng.uniform(-0.3, 0.3, p), robust=True, eps_max=0.002, lamb_alpha=20.0, lamb_sigma=500.0, c=1.0, slack=slack, tec=True)
    B = 3
    d = generate_batch(range(B), N=N, plant=plant)
    up = d["u_d"][:, -n:, :].reshape(B, -1).copy(); yp = d["y_d"][:, -n:, :].reshape(B, -1).copy()
    res = {}
    for mode in (L.GRAM_DENSE, L.GRAM_STRUCTURED):
        with BatchedDDMPC(n=n, m=m, p=p, L_=Lh, N=N, Q=2.0, R=0.05, u_s=spec.u_s, y_s=spec.y_s, batch=B, controller_type=L.ROBUST,
                          slack_type=L.SLACK_CONVEX if slack == "convex" else L.SLACK_NONE, eps_max=0.002, lamb_alpha=20.0,
                          lamb_sigma=500.0, c=1.0, gram_mode=mode) as eng:
            assert "cold" in eng.kernel_name()
            eng.set_refinement("always")
            eng.set_data(d["u_d"], d["y_d"])
            res[mode] = tuple(x.copy() for x in eng.solve(up, yp))
            uw = eng.step(up, yp)
            assert np.max(np.abs(uw[0] - res[mode][0])) <= 1e-8 * np.max(np.abs(res[mode][0])) and np.array_equal(uw[2], res[mode][2])
    dn, st = res[L.GRAM_DENSE], res[L.GRAM_STRUCTURED]
    for b in range(B):
        sol = orc.solve_fullspace(spec, d["u_d"][b], d["y_d"][b], up[b], yp[b])
        for r_ in (dn, st):
            assert L.STATUS_STRINGS[int(r_[2][b])] == sol.status == "optimal"
            assert np.max(np.abs(r_[0][b] - sol.optimal_u)) / max(np.max(np.abs(sol.optimal_u)), 1e-3) < 1e-8, (rows, b)
            assert abs(r_[1][b] - sol.cost) <= 1e-8 * max(abs(sol.cost), 1e-6), (rows, b)
    assert np.max(np.abs(dn[0] - st[0])) <= 1e-9 * np.max(np.abs(dn[0])) and np.array_equal(dn[3], st[3])


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", ["cfg2", "cfg4"])
@pytest.mark.parametrize("slack", ["none", "convex"])
def test_headline_config_against_the_extended_precision_golden_solutions(slack, cfg):
    """BASELINE configs[1] (four-tank robust DD-MPC, L = 30, N = 400) against the QP as the reference states it, solved in 80-bit
    arithmetic (tests/golden/cfg2_extended.npz, make_golden_cfg2_extended.py; the fp64 checkers are pinned to it on the CPU in
    test_oracle.py): cold solve in every refinement mode and the warm step, six instances, slack NONE and CONVEX, at 1e-10 -- two
    orders inside the suite's bars; CONVEX: the same active-set iteration counts."""
    from direct_data_driven_mpc_amd.harness import generate_batch
    # (cfg4: BASELINE configs[3], L = 60, N = 1000 on the <17,8> instance, two instances, tests/golden/cfg4_extended.npz)
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", cfg + "_extended.npz"))
    seeds = [int(s) for s in z["seeds"]]
    B = len(seeds)
    Lh, N = int(z["L"]), int(z["N"])
    spec = orc.spec_from_params(L=Lh, N=N, **({"slack_var_constraint_type": 1} if slack == "convex" else {}))
    n = spec.n
    d = generate_batch(seeds, N=N)
    up = d["u_d"][:, -n:, :].reshape(B, -1).copy(); yp = d["y_d"][:, -n:, :].reshape(B, -1).copy()
    ug, cg = z["optimal_u_" + slack], z["cost_" + slack]
    sc = np.max(np.abs(ug), axis=1, keepdims=True)
    for refine in ("auto", "off", "always"):
        with _spec_engine(spec, N, B) as eng:
            eng.set_refinement(refine)
            eng.set_data(d["u_d"], d["y_d"])
            u, cost, status, iters = eng.solve(up, yp)
            uw, cw, sw, iw = eng.step(up, yp)
        assert np.all(status == 0) and np.all(sw == 0)
        assert np.max(np.abs(u - ug) / sc) < 1e-10 and np.max(np.abs(cost - cg) / cg) < 1e-10, refine
        assert np.max(np.abs(uw - ug) / sc) < 1e-10 and np.max(np.abs(cw - cg) / cg) < 1e-10, refine
        if slack == "convex":
            assert np.array_equal(iters, z["iters_convex"]) and np.array_equal(iw, iters)
