"""GPU tests added in round 2: the device paths against fixtures produced by the REFERENCE's own driver code
(tests/golden/reference_loops.npz, see tests/golden/make_golden_loops.py), BASELINE configs[0] end to end through
the class, a configs[2] rank shard, and the multi-rank bench path with the real engine."""
import io
import json
import os
import re
import subprocess
import sys
from contextlib import redirect_stdout

import numpy as np
import pytest

from direct_data_driven_mpc_amd import _lib as L
from direct_data_driven_mpc_amd import harness
from direct_data_driven_mpc_amd.engine import BatchedDDMPC
from oracle import ddmpc_oracle as orc
from oracle import oracle_c

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL_U, TOL_COST = 1e-8, 1e-9


@pytest.fixture(scope="module")
def loops():
    return np.load(os.path.join(ROOT, "tests", "golden", "reference_loops.npz"))


def _engine(cfg, B, tec=True):
    return BatchedDDMPC(n=cfg["n"], m=cfg["m"], p=cfg["p"], L_=cfg["L"], N=cfg["N"], Q=cfg["Q"], R=cfg["R"], u_s=cfg["u_s"],
                        y_s=cfg["y_s"], batch=B, controller_type=L.ROBUST if cfg["robust"] else L.NOMINAL,
                        slack_type=L.SLACK_CONVEX if cfg["slack"] == "convex" else L.SLACK_NONE, eps_max=cfg["eps_max"],
                        lamb_alpha=cfg["lamb_alpha"], lamb_sigma=cfg["lamb_sigma"], c=cfg["c"], use_terminal_constraint=tec)


# ------------------------------------------------------------------ f1: device closed loop vs the reference's loop code
@pytest.mark.parametrize("tag,seed,over,step", [("ex_robust_s0", 0, {}, 4), ("ex_robust_s4", 4, {}, 4),
                                                ("ex_convex1_s0", 0, dict(slack_var_constraint_type=1), 1),
                                                ("ex_nominal_s0", 0, dict(controller_type=0), 4)])
@pytest.mark.parametrize("path", ["auto", "cold"])
def test_device_closed_loop_equals_reference_run(gpu, loops, tag, seed, over, step, path):
    u_ref, y_ref = loops[tag + "_u_sys"], loops[tag + "_y_sys"]
    n_steps = u_ref.shape[0]
    cfg = harness.controller_params(over)
    d = harness.generate_batch([seed, seed + 100])            # instance 0 = the reference run, 1 = a bystander
    P = harness.FOUR_TANK
    w = np.stack([P["eps_max"] * rng.uniform(-1.0, 1.0, (n_steps, 2)) for rng in d["rngs"]])
    n = cfg["n"]
    with _engine(cfg, 2) as eng:
        eng.set_data(d["u_d"], d["y_d"])
        eng.set_closed_loop_path(path)
        u_sys, y_sys, st, *_ = eng.closed_loop(P["A"], P["B"], P["C"], P["D"], d["x_end"], d["u_d"][:, -n:].reshape(2, -1),
                                               d["y_d"][:, -n:].reshape(2, -1), w, n_mpc_step=step)
    assert np.all(st == 0)
    assert np.max(np.abs(u_sys[0] - u_ref)) / np.max(np.abs(u_ref)) < TOL_U
    assert np.max(np.abs(y_sys[0] - y_ref)) < 1e-9


# ------------------------------------------------------------------ cfg 1: one nominal controller through the class
def test_config1_single_nominal_controller_end_to_end(gpu, loops):
    # BASELINE configs[0]: the reference example with --controller_type Nominal --seed 0 --t_sim 400, the controller
    # created as controller_creation.py:255-273 does and driven as controller_operation.py:269-305 does
    from direct_data_driven_mpc.direct_data_driven_mpc_controller import (DataDrivenMPCType, DirectDataDrivenMPCController,
                                                                          SlackVarConstraintTypes)
    cfg = harness.controller_params(dict(controller_type=0))
    d = harness.generate_batch([0])
    ctrl = DirectDataDrivenMPCController(
        n=cfg["n"], m=cfg["m"], p=cfg["p"], u_d=d["u_d"][0], y_d=d["y_d"][0], L=cfg["L"], Q=cfg["Q"] * np.eye(cfg["p"] * cfg["L"]),
        R=cfg["R"] * np.eye(cfg["m"] * cfg["L"]), u_s=cfg["u_s"].reshape(-1, 1), y_s=cfg["y_s"].reshape(-1, 1),
        eps_max=cfg["eps_max"], lamb_alpha=cfg["lamb_alpha"], lamb_sigma=cfg["lamb_sigma"], c=cfg["c"],
        slack_var_constraint_type=SlackVarConstraintTypes.NONE, controller_type=DataDrivenMPCType.NOMINAL,
        n_mpc_step=cfg["n_mpc_step"], use_terminal_constraint=True)
    assert ctrl.get_problem_solve_status() == "optimal"
    buf = io.StringIO()
    with redirect_stdout(buf):
        u_sys, y_sys, _ = harness.simulate_control_loop(harness.FOUR_TANK, d["x_end"][0], ctrl, 401, d["rngs"][0], verbose=2)
    assert u_sys.shape == (401, 2) and np.max(np.abs(u_sys - cfg["u_s"])) < 1e-8          # u == u_s at EVERY step
    assert np.max(np.abs(u_sys - loops["ex_nominal_s0_u_sys"])) < 1e-8
    assert np.max(np.abs(y_sys - loops["ex_nominal_s0_y_sys"])) < 1e-9
    lines = buf.getvalue().splitlines()
    ref_lines = list(loops["ex_nominal_s0_lines"])
    assert len(lines) == len(ref_lines) == 101                                               # one line per solve, t = 0, 4, ..., 400
    unsign = lambda s: re.sub(r"-(0\.0+)(?![0-9])", r" \1", s)                               # -0.0000 vs 0.0000: rounding noise
    assert [unsign(a) for a in lines] == [unsign(b) for b in ref_lines]


# ------------------------------------------------------------------ f3: the example scripts vs the reference runs
def test_example_scripts_equal_reference_runs(gpu, loops, tmp_path):
    out = tmp_path / "ex.npz"
    res = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "batched_data_driven_mpc_example.py"), "--batch", "5",
                          "--seed", "0", "--t_sim", "400", "--verbose", "2", "--out", str(out)],
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    z = np.load(out)
    for inst, tag in ((0, "ex_robust_s0"), (4, "ex_robust_s4")):
        assert np.max(np.abs(z["u_sys"][inst] - loops[tag + "_u_sys"])) / np.max(np.abs(loops[tag + "_u_sys"])) < TOL_U
        assert np.max(np.abs(z["y_sys"][inst] - loops[tag + "_y_sys"])) < 1e-9
    got = [ln for ln in res.stdout.splitlines() if "Time step" in ln]
    assert got == list(loops["ex_robust_s0_lines"])                                          # instance 0 = --seed 0 of the reference
    out = tmp_path / "rep.npz"
    res = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "batched_robust_reproduction.py"), "--batch", "5",
                          "--seed", "0", "--t_sim", "600", "--out", str(out)], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    z = np.load(out)
    for inst in (0, 4):
        for key, tag in (("TEC_1-step", "tec"), ("TEC_n-step", "tec_nstep"), ("UCON_1-step", "ucon")):
            u_ref, y_ref = loops["rep_s%d_%s_u" % (inst, tag)], loops["rep_s%d_%s_y" % (inst, tag)]
            assert z[key + "_u"].shape[1:] == u_ref.shape == (597, 2)
            # UCON diverges (the paper's point; |u| reaches 1e3 and more): errors are measured against the size the
            # trajectory has reached at that step
            scale = np.maximum(1.0, np.maximum.accumulate(np.max(np.abs(np.hstack([u_ref, y_ref])), axis=1)))[:, None]
            assert np.max(np.abs(z[key + "_u"][inst] - u_ref) / scale) < 1e-7, (inst, tag)
            assert np.max(np.abs(z[key + "_y"][inst] - y_ref) / scale) < 1e-8, (inst, tag)


# ------------------------------------------------------------------ cfg 3: one rank's shard of 262,144
def test_config3_rank_shard(gpu):
    # BASELINE configs[2] = 262,144 instances over 8 GPUs: rank 5's shard (32,768 instances, seeds 163,840...),
    # every instance optimal and checked against the compiled CPU restatement, a sample against the full-space
    # numpy oracle, and results independent of which other instances share the batch
    from direct_data_driven_mpc_amd.distributed import shard_bounds
    lo, hi = shard_bounds(262144, 5, 8)
    assert (lo, hi) == (163840, 196608)
    B = hi - lo
    cfg = harness.controller_params()
    d = harness.generate_batch(range(lo, hi))
    n = cfg["n"]
    up = d["u_d"][:, -n:, :].reshape(B, -1).copy(); yp = d["y_d"][:, -n:, :].reshape(B, -1).copy()
    with _engine(cfg, B) as eng:
        eng.set_data(d["u_d"], d["y_d"])
        u, cost, status, iters = eng.solve(up, yp)
    assert np.all(status == 0) and np.all(iters == 1)
    spec = orc.spec_from_params()
    uc, cc, sc, _ = oracle_c.solve_batch(spec, cfg["N"], d["u_d"], d["y_d"], up, yp, threads=min(16, os.cpu_count() or 1))
    assert np.all(sc == 0)
    assert np.max(np.max(np.abs(u - uc), axis=1) / np.max(np.abs(uc), axis=1)) < TOL_U
    assert np.max(np.abs(cost - cc) / np.abs(cc)) < TOL_COST
    for b in range(0, B, 4099):
        sol = orc.solve_fullspace(spec, d["u_d"][b], d["y_d"][b], up[b], yp[b])
        assert np.max(np.abs(u[b] - sol.optimal_u)) / np.max(np.abs(sol.optimal_u)) < TOL_U
        assert abs(cost[b] - sol.cost) / abs(sol.cost) < TOL_COST
    pick = np.random.default_rng(0).choice(B, 1000, replace=False)
    with _engine(cfg, pick.size) as eng:
        eng.set_data(d["u_d"][pick], d["y_d"][pick])
        u2, c2, s2, _ = eng.solve(up[pick], yp[pick])
    assert np.array_equal(u2, u[pick]) and np.array_equal(c2, cost[pick]) and np.array_equal(s2, status[pick])


# ------------------------------------------------------------------ e: the multi-rank bench path with the real engine
def test_bench_self_launches_two_ranks_and_gathers_bit_exactly(gpu, tmp_path):
    # `python bench.py --gpus 2` from a bare shell (no WORLD_SIZE): bench.py starts its own two ranks; on this one-GPU
    # box they share device 0 and gather over gloo (--rehearse-on-one-gpu).  The gathered block must equal the
    # single-process solve of the same 2 x 96 instances bit for bit.
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    g2, g1 = tmp_path / "g2.npz", tmp_path / "g1.npz"
    common = ["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-warm"]
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-on-one-gpu",
                          "--batch-per-gpu", "96", "--dump-gathered", str(g2)] + common,
                         capture_output=True, text=True, timeout=900, env=env)
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-3000:])
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["config"]["global_batch"] == 192 and rec["config"]["non_optimal_instances"] == 0
    # what the process group itself saw (round 4): two ranks, gloo in this rehearsal, the device each one bound
    dinfo = rec["distributed"]
    assert dinfo["world_size"] == 2 and dinfo["backend"] == "gloo" and [r["rank"] for r in dinfo["ranks"]] == [0, 1]
    assert all(r["device"] == 0 and r["name"] for r in dinfo["ranks"])
    # round 5: the N > 1 line carries the single-GPU rate at the SAME per-GPU batch and the efficiency against it
    ref = rec["per_gpu_reference"]
    assert ref["batch"] == 96 and ref["steps"] == 2 and ref["value"] > 0 and ref["ms_per_step"] > 0
    assert abs(rec["scaling_efficiency_vs_same_batch"] - rec["value"] / (2 * ref["value"])) < 1e-12
    assert "configs[2]" in rec["config"]["workload"] or "custom batch" in rec["config"]["workload"]
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--batch-per-gpu", "192",
                          "--dump-gathered", str(g1)] + common, capture_output=True, text=True, timeout=900, env=env)
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-3000:])
    a, b = np.load(g2), np.load(g1)
    assert a["u"].shape == (192, 60)
    assert np.array_equal(a["u"], b["u"]) and np.array_equal(a["cost"], b["cost"]) and np.array_equal(a["status"], b["status"])
    # a world size that contradicts --gpus is an error, not silently ignored
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + common, capture_output=True, text=True,
                         timeout=300, env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert bad.returncode != 0 and "does not match WORLD_SIZE" in (bad.stderr + bad.stdout)


def test_bench_rccl_branch_at_world_size_one(gpu, tmp_path):
    # The branch an N-GPU run takes -- init_process_group("nccl", device_id=...), barrier(device_ids=...), the packed
    # all_gather_into_tensor on DEVICE tensors (RCCL), all_reduce(MAX) of the step time -- executed on this one-GPU box:
    # `bench.py --gpus 1 --force-dist` starts one rank through torch.distributed.run before anything touches the GPU.
    # Its gathered block must equal the plain single-process run bit for bit.
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    gd, gp = tmp_path / "gd.npz", tmp_path / "gp.npz"
    common = ["--gpus", "1", "--batch-per-gpu", "160", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-warm"]
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--dump-gathered", str(gd)] + common,
                         capture_output=True, text=True, timeout=900, env=env)
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-3000:])
    rec = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["n_gpus"] == 1 and rec["config"]["global_batch"] == 160 and rec["config"]["non_optimal_instances"] == 0
    assert "RCCL branch forced" in rec["config"]["parallelism"]
    assert rec["distributed"]["world_size"] == 1 and rec["distributed"]["backend"] == "nccl" and rec["distributed"]["ranks"][0]["device"] == 0
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dump-gathered", str(gp)] + common,
                         capture_output=True, text=True, timeout=900, env=env)
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-3000:])
    a, b = np.load(gd), np.load(gp)
    assert a["u"].shape == (160, 60)
    assert np.array_equal(a["u"], b["u"]) and np.array_equal(a["cost"], b["cost"]) and np.array_equal(a["status"], b["status"])


# ------------------------------------------------------------------ nominal rescue behind every entry point
def _exact_nominal(B=6):
    from direct_data_driven_mpc_amd.harness import FOUR_TANK, generate_batch
    plant = dict(FOUR_TANK); plant["eps_max"] = 0.0
    d = generate_batch(range(B), N=400, plant=plant)
    A, Bm, Cm, D = (FOUR_TANK[k] for k in "ABCD")
    spec = orc.spec_from_params(controller_type=0)
    spec.y_s = (Cm @ np.linalg.inv(np.eye(4) - A) @ Bm + D) @ spec.u_s          # a true equilibrium: the nominal QP is feasible
    up = d["u_d"][:, -4:, :].reshape(B, -1).copy(); yp = d["y_d"][:, -4:, :].reshape(B, -1).copy()
    return spec, d, up, yp


def _spec_engine(spec, N, B):
    return BatchedDDMPC(n=spec.n, m=spec.m, p=spec.p, L_=spec.L, N=N, Q=spec.Q, R=spec.R, u_s=spec.u_s, y_s=spec.y_s, batch=B,
                        controller_type=L.ROBUST if spec.robust else L.NOMINAL,
                        slack_type=L.SLACK_CONVEX if spec.slack == "convex" else L.SLACK_NONE, eps_max=spec.eps_max,
                        lamb_alpha=spec.lamb_alpha, lamb_sigma=spec.lamb_sigma, c=spec.c, use_terminal_constraint=spec.tec)


def test_pipelined_host_solve_runs_the_nominal_rescue(gpu):
    # ddmpc_solve_from_host must give what ddmpc_set_data + ddmpc_solve give, also for a NOMINAL controller on exact
    # (rank-deficient) data, where every instance is solved by the rank-revealing rescue kernel
    B = 300                                              # > 256: four chunks
    spec, d, up, yp = _exact_nominal(B)
    with _spec_engine(spec, 400, B) as eng:
        u1, c1, s1, i1 = eng.solve_from_host(d["u_d"], d["y_d"], up, yp)
        eng.set_data(d["u_d"], d["y_d"])
        u2, c2, s2, i2 = eng.solve(up, yp)
    assert np.all(s2 == 0)
    assert np.array_equal(u1, u2) and np.array_equal(c1, c2) and np.array_equal(s1, s2) and np.array_equal(i1, i2)


def test_variables_after_a_rescued_nominal_solve(gpu):
    # .ubar/.ybar after an exact-data nominal solve come from the rescue kernel's own z (not from the failed fast
    # path's workspace); alpha = H' x from the vector x the kernel exports (z = H H' x): H alpha reproduces [ubar; ybar]
    from oracle.nominal_exact import solve_nominal_exact
    B = 4
    spec, d, up, yp = _exact_nominal(B)
    n, m, p, Lh = spec.n, spec.m, spec.p, spec.L
    with _spec_engine(spec, 400, B) as eng:
        eng.set_data(d["u_d"], d["y_d"])
        u, cost, status, _ = eng.solve(up, yp)
        ub, yb, al = eng.get_solution("ubar"), eng.get_solution("ybar"), eng.get_solution("alpha")
        uw, _, sw, _ = eng.step(up, yp)
        ub2 = eng.get_solution("ubar")
    assert np.all(status == 0) and np.all(sw == 0)
    assert np.array_equal(ub[:, n * m:], u) and np.array_equal(ub2[:, n * m:], uw)
    assert np.array_equal(ub[:, :n * m], up) and np.array_equal(yb[:, :n * p], yp)             # internal-state constraint
    assert np.allclose(yb[:, Lh * p:], np.tile(spec.y_s, n), atol=1e-12)                      # terminal constraint
    assert al.shape == (B, 400 - spec.Ln + 1) and np.all(np.isfinite(al))
    for b in range(B):
        ref = solve_nominal_exact(spec, d["u_d"][b], d["y_d"][b], up[b], yp[b])
        assert np.max(np.abs(ub[b, n * m:] - ref["optimal_u"])) / np.max(np.abs(ref["optimal_u"])) < 1e-8
        Hu, Hy = orc.hankel_matrix(d["u_d"][b], spec.Ln), orc.hankel_matrix(d["y_d"][b], spec.Ln)
        sc = max(np.max(np.abs(ub[b])), np.max(np.abs(yb[b])))
        assert np.max(np.abs(Hu @ al[b] - ub[b])) < 1e-7 * sc and np.max(np.abs(Hy @ al[b] - yb[b])) < 1e-7 * sc
    # a noisy (full-rank) nominal batch right afterwards on the same handle: fast path, alpha is real again
    dn = harness.generate_batch(range(B))
    with _spec_engine(orc.spec_from_params(controller_type=0), 400, B) as eng:
        eng.set_data(dn["u_d"], dn["y_d"])
        eng.solve(dn["u_d"][:, -4:, :].reshape(B, -1), dn["y_d"][:, -4:, :].reshape(B, -1))
        assert np.all(np.isfinite(eng.get_solution("alpha")))


def test_set_stream_is_idempotent_and_orders_streams(gpu):
    torch = pytest.importorskip("torch")
    B = 8
    cfg = harness.controller_params()
    d = harness.generate_batch(range(B))
    dev = torch.device("cuda", 0)
    ud, yd = torch.from_numpy(d["u_d"]).to(dev), torch.from_numpy(d["y_d"]).to(dev)
    up = torch.from_numpy(d["u_d"][:, -4:, :].reshape(B, -1).copy()).to(dev)
    yp = torch.from_numpy(d["y_d"][:, -4:, :].reshape(B, -1).copy()).to(dev)
    with _engine(cfg, B) as eng:
        eng.set_data(ud, yd)
        ref = [t.clone() for t in eng.solve(up, yp)]
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):                       # a non-default torch stream, used for several calls in a row
            outs = [eng.solve(up, yp) for _ in range(3)]
        s.synchronize()
        for o in outs:
            assert all(torch.equal(a, b) for a, b in zip(o, ref))


# ------------------------------------------------------------------ refinement with exact Hankel products (cold kernel)
def test_auto_refinement_flags_ill_conditioned_instances_only(gpu):
    # default (DDMPC_REFINE_AUTO): the plain kernel checks every solve with the exact-Hankel residual, flags the instances
    # above the threshold and only those are solved again by the refining variant.  Benchmark data: nothing flagged, results
    # bit-equal to refinement OFF.  An ill-conditioned plant (random, high output gain against the noise level): AUTO
    # equals ALWAYS and meets the standard bars, OFF does not.
    import test_gpu_parity as T
    spec = orc.spec_from_params()
    B = 16
    d = harness.generate_batch(range(B))
    up = d["u_d"][:, -4:, :].reshape(B, -1).copy(); yp = d["y_d"][:, -4:, :].reshape(B, -1).copy()
    with _spec_engine(spec, 400, B) as eng:
        eng.set_data(d["u_d"], d["y_d"])
        eng.set_refinement("off"); u0, c0, s0, _ = (x.copy() for x in eng.solve(up, yp))
        eng.set_refinement("auto"); u1, c1, s1, _ = (x.copy() for x in eng.solve(up, yp))
        eng.set_refinement("always"); u2, c2, s2, _ = (x.copy() for x in eng.solve(up, yp))
    assert np.array_equal(u0, u1) and np.array_equal(c0, c1)
    assert np.max(np.abs(u2 - u0)) / np.max(np.abs(u0)) < 1e-10           # well-conditioned: refinement changes nothing visible
    rng = np.random.default_rng(1017)                                       # case 17 of the random-plant sweep: cond(K) ~ 2e7
    m, p, ns = 2, 3, 4
    plant = T._random_plant(np.random.default_rng(1017), ns, m, p, 0.002)
    Lh, N = 16, 200
    spec = orc.QPSpec(n=ns, m=m, p=p, L=Lh, Q=2.0 * np.eye(p * Lh), R=0.05 * np.eye(m * Lh), u_s=rng.uniform(-0.5, 0.5, m),
                      y_s=rng.uniform(-0.5, 0.5, p), robust=True, eps_max=0.002, lamb_alpha=20.0, lamb_sigma=500.0, c=1.0,
                      slack="none", tec=True)
    B = 6
    d = harness.generate_batch(range(170, 170 + B), N=N, plant=plant)
    up = d["u_d"][:, -ns:, :].reshape(B, -1).copy(); yp = d["y_d"][:, -ns:, :].reshape(B, -1).copy()
    out = {}
    with _spec_engine(spec, N, B) as eng:
        eng.set_data(d["u_d"], d["y_d"])
        warm = {}
        for mode in ("off", "auto", "always"):
            eng.set_refinement(mode)                                         # (also invalidates the affine law of the warm step)
            out[mode] = tuple(x.copy() for x in eng.solve(up, yp))
            warm[mode] = eng.step(up, yp)[0].copy()
    err = {}
    for mode, (u, c, s, _) in out.items():
        eu = ec = 0.0
        for b in range(B):
            sol = orc.solve_fullspace(spec, d["u_d"][b], d["y_d"][b], up[b], yp[b])
            eu = max(eu, np.max(np.abs(u[b] - sol.optimal_u)) / np.max(np.abs(sol.optimal_u)))
            ec = max(ec, abs(c[b] - sol.cost) / abs(sol.cost))
        err[mode] = (eu, ec)
    assert err["always"][0] < TOL_U and err["always"][1] < TOL_COST, err
    assert err["auto"][0] < TOL_U and err["auto"][1] < TOL_COST, err
    assert err["off"][0] > 10 * err["always"][0], err                          # the Gram route alone is visibly worse here
    # the warm step: its affine law comes from refining solves for the instances AUTO flags (ddmpc_prepare), so it meets the
    # cold solve at the same bar; with refinement off it inherits the Gram route's error
    scale = np.max(np.abs(out["always"][0]))
    assert np.max(np.abs(warm["always"] - out["always"][0])) < TOL_U * scale
    assert np.max(np.abs(warm["auto"] - out["always"][0])) < TOL_U * scale
    assert np.max(np.abs(warm["off"] - out["always"][0])) > 10 * np.max(np.abs(warm["auto"] - out["always"][0]))


@pytest.mark.gpu
def test_auto_refinement_is_decided_per_solve_and_borrowed_data_may_change_in_place(gpu):
    # AUTO keeps nothing about a data set: every solve is checked with its own exact-Hankel residual.  Device trajectories are
    # borrowed (include/ddmpc.h, ddmpc_set_data): rewriting them in place between two ddmpc_solve calls -- benign data first,
    # then data whose Gram route misses the bars -- must give exactly what a fresh handle gives on the new contents.
    import torch
    import test_gpu_parity as T
    rng = np.random.default_rng(1017)
    m, p, ns = 2, 3, 4
    plant = T._random_plant(np.random.default_rng(1017), ns, m, p, 0.002)      # ill-conditioned: every instance gets flagged
    benign = dict(plant); benign["C"] = 0.02 * plant["C"]; benign["eps_max"] = 0.002   # output gain ~ noise level: cond(H) small
    Lh, N, B = 16, 200, 6
    spec = orc.QPSpec(n=ns, m=m, p=p, L=Lh, Q=2.0 * np.eye(p * Lh), R=0.05 * np.eye(m * Lh), u_s=rng.uniform(-0.5, 0.5, m),
                      y_s=rng.uniform(-0.5, 0.5, p), robust=True, eps_max=0.002, lamb_alpha=20.0, lamb_sigma=500.0, c=1.0,
                      slack="none", tec=True)
    d_ill = harness.generate_batch(range(170, 170 + B), N=N, plant=plant)
    d_ok = harness.generate_batch(range(270, 270 + B), N=N, plant=benign)
    dev = torch.device("cuda", 0)

    def past(d):
        return d["u_d"][:, -ns:, :].reshape(B, -1).copy(), d["y_d"][:, -ns:, :].reshape(B, -1).copy()

    def fresh(d, mode):
        with _spec_engine(spec, N, B) as e2:
            e2.set_refinement(mode)
            e2.set_data(d["u_d"], d["y_d"])
            return tuple(x.copy() for x in e2.solve(*past(d)))

    ud = torch.from_numpy(d_ok["u_d"]).to(dev); yd = torch.from_numpy(d_ok["y_d"]).to(dev)
    with _spec_engine(spec, N, B) as eng:
        eng.set_data(ud, yd)                                            # borrowed device tensors
        r_ok = tuple(x.copy() for x in eng.solve(*past(d_ok)))
        off_ok = fresh(d_ok, "off")
        assert np.array_equal(r_ok[0], off_ok[0]) and np.array_equal(r_ok[1], off_ok[1])        # benign data: nothing flagged
        ud.copy_(torch.from_numpy(d_ill["u_d"])); yd.copy_(torch.from_numpy(d_ill["y_d"]))      # in place, no ddmpc_set_data
        torch.cuda.synchronize()
        r_ill = tuple(x.copy() for x in eng.solve(*past(d_ill)))
        again = tuple(x.copy() for x in eng.solve(*past(d_ill)))
    auto_ill, always_ill, off_ill = fresh(d_ill, "auto"), fresh(d_ill, "always"), fresh(d_ill, "off")
    assert np.array_equal(r_ill[0], auto_ill[0]) and np.array_equal(r_ill[1], auto_ill[1]) and np.array_equal(r_ill[2], auto_ill[2])
    assert np.array_equal(r_ill[0], again[0]) and np.array_equal(r_ill[1], again[1])
    scale = np.max(np.abs(always_ill[0]))
    assert np.max(np.abs(r_ill[0] - always_ill[0])) <= 1e-12 * scale                            # the flagged instances were refined
    assert np.max(np.abs(off_ill[0] - always_ill[0])) > 1e-10 * scale                           # ... and needed it
    for b in range(B):
        sol = orc.solve_fullspace(spec, d_ill["u_d"][b], d_ill["y_d"][b], past(d_ill)[0][b], past(d_ill)[1][b])
        assert np.max(np.abs(r_ill[0][b] - sol.optimal_u)) / np.max(np.abs(sol.optimal_u)) < TOL_U
        assert abs(r_ill[1][b] - sol.cost) <= TOL_COST * abs(sol.cost)


@pytest.mark.gpu
@pytest.mark.parametrize("case", [1, 2, 4], ids=["m3p2-convex", "m1p4-none", "m2p5-none-ucon"])
def test_global_workspace_kernels_with_channel_counts_that_do_not_fill_a_tile(gpu, case):
    # beyond 271 rows the Gram, the Cholesky, the Schur complement and C'WC run on 16x16 MFMA tiles over packed matrices:
    # channel counts 5 and 7 (partly filled lag tiles, panels that end inside a tile) on seeded random plants, against the
    # full-space oracle (the cases of tools/large_fuzz.py; controller.py:506-547,679-722)
    from direct_data_driven_mpc_amd.harness import generate_batch
    rng = np.random.default_rng(5000 + case)
    m, p = [(2, 2), (3, 2), (1, 4), (4, 1), (2, 5), (3, 3)][case % 6]
    ns = n = int(rng.integers(2, 5))
    rows = int(rng.integers(280, 780))
    Lh = max(2 * n, rows // (m + p) - n)
    N = (m + 1) * (Lh + 2 * n) + int(rng.integers(150, 400))
    eps = 0.002
    slack = "convex" if case % 2 == 1 else "none"
    tec = case % 5 != 4
    Q = np.diag(rng.uniform(1.0, 4.0, p * Lh)); R = np.diag(rng.uniform(0.01, 0.1, m * Lh))
    A = rng.normal(size=(ns, ns)); A *= rng.uniform(0.5, 0.9) / max(abs(np.linalg.eigvals(A)))
    plant = dict(A=A, B=rng.normal(size=(ns, m)), C=rng.normal(size=(p, ns)), D=np.zeros((p, m)), eps_max=eps)
    spec = orc.QPSpec(n=n, m=m, p=p, L=Lh, Q=Q, R=R, u_s=rng.uniform(-0.5, 0.5, m), y_s=rng.uniform(-0.5, 0.5, p),
                      robust=True, eps_max=eps, lamb_alpha=20.0, lamb_sigma=500.0, c=1.0, slack=slack, tec=tec)
    B = 2
    d = generate_batch(range(case * 10, case * 10 + B), N=N, plant=plant)
    up = d["u_d"][:, -n:, :].reshape(B, -1).copy(); yp = d["y_d"][:, -n:, :].reshape(B, -1).copy()
    with BatchedDDMPC(n=n, m=m, p=p, L_=Lh, N=N, Q=np.diag(Q), R=np.diag(R), u_s=spec.u_s, y_s=spec.y_s, batch=B,
                      controller_type=L.ROBUST, slack_type=L.SLACK_CONVEX if slack == "convex" else L.SLACK_NONE, eps_max=eps,
                      lamb_alpha=20.0, lamb_sigma=500.0, c=1.0, use_terminal_constraint=tec) as eng:
        assert eng.kernel_name() == "ddmpc_large_solve_kernel" and (m + p) * (Lh + n) > 271
        eng.set_data(d["u_d"], d["y_d"])
        u, cost, status, iters = eng.solve(up, yp)
    for b in range(B):
        sol = orc.solve_fullspace(spec, d["u_d"][b], d["y_d"][b], up[b], yp[b])
        assert L.STATUS_STRINGS[int(status[b])] == sol.status == "optimal"
        if slack == "convex":
            assert int(iters[b]) == sol.iters
        assert np.max(np.abs(u[b] - sol.optimal_u)) <= TOL_U * max(np.max(np.abs(sol.optimal_u)), 1e-3)
        assert abs(cost[b] - sol.cost) <= TOL_COST * max(abs(sol.cost), 1e-6)


@pytest.mark.gpu
def test_nominal_rescue_on_exact_data_of_a_plant_with_five_channels(gpu):
    # the rank-revealing rescue (exact data, controller.py:506-538) on a random stable plant with m = 2, p = 3: the lag
    # tiles of its Hankel Gram are only partly filled; against the SVD-based CPU solve and the analytic properties
    from direct_data_driven_mpc_amd.harness import generate_batch
    from oracle.nominal_exact import solve_nominal_exact
    rng = np.random.default_rng(77)
    ns = n = 3; m, p = 2, 3; Lh = 12; N = 260; B = 3
    A = rng.normal(size=(ns, ns)); A *= 0.8 / max(abs(np.linalg.eigvals(A)))
    plant = dict(A=A, B=rng.normal(size=(ns, m)), C=rng.normal(size=(p, ns)), D=np.zeros((p, m)), eps_max=0.0)
    u_s = np.array([0.3, -0.2]); y_s = (plant["C"] @ np.linalg.inv(np.eye(ns) - A) @ plant["B"]) @ u_s    # a true equilibrium
    spec = orc.QPSpec(n=n, m=m, p=p, L=Lh, Q=2.0 * np.eye(p * Lh), R=0.1 * np.eye(m * Lh), u_s=u_s, y_s=y_s, robust=False,
                      eps_max=0.0, lamb_alpha=0.0, lamb_sigma=0.0, c=0.0, slack="none", tec=True)
    d = generate_batch(range(B), N=N, plant=plant)
    up = d["u_d"][:, -n:, :].reshape(B, -1).copy(); yp = d["y_d"][:, -n:, :].reshape(B, -1).copy()
    with BatchedDDMPC(n=n, m=m, p=p, L_=Lh, N=N, Q=2.0, R=0.1, u_s=u_s, y_s=y_s, batch=B, controller_type=L.NOMINAL) as eng:
        eng.set_data(d["u_d"], d["y_d"])
        u, cost, status, _ = eng.solve(up, yp)
    for b in range(B):
        ref = solve_nominal_exact(spec, d["u_d"][b], d["y_d"][b], up[b], yp[b])
        assert ref["status"] == "optimal" and L.STATUS_STRINGS[int(status[b])] == "optimal", (b, ref["residual"])
        assert np.max(np.abs(u[b] - ref["optimal_u"])) <= TOL_U * np.max(np.abs(ref["optimal_u"])), b
        assert abs(cost[b] - ref["cost"]) <= 1e-8 * max(abs(ref["cost"]), 1e-6)
