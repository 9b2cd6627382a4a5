#!/usr/bin/env python
"""bench.py -- QP solves/sec of the batched Data-Driven MPC cold solve on MI355X.

Workload (BASELINE.json configs[1]): four-tank robust DD-MPC, L=30, N=400, slack
NONE (the reference YAML default), batch = 4096 noise seeds per GPU, inputs
resident in HBM.  One "step" = one cold QP solve (implicit Hankel -> Gram ->
reduced KKT -> Cholesky -> solve) for every instance of the batch.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.  See DESIGN.md "Measurement".
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6   # 256 CU x 4 SIMD x 2.4 GHz x 2048 flop / 64 clk (v_mfma_f64_16x16x4_f64),
                               # measured 64 clk/instr/SIMD in profiles/r01_mfma_f64_probe.log
# HBM traffic of one default launch (4096 instances, slack NONE, structured Gram) from separate
# rocprofv3 --pmc passes of this same command (profiles/r01_final_pmc_fetch.csv / _write.csv):
# FETCH_SIZE 28,911 KB x2 (gfx950 counts wide reads at half) + WRITE_SIZE 13,050 KB.
HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
PMC_TRAFFIC_BYTES_DEFAULT = (2 * 28911 + 13050) * 1024
# the same for ddmpc_warm_step_kernel at 4096 instances (profiles/r01_warm_step_pmc_*.csv)
WARM_PMC_TRAFFIC_BYTES_DEFAULT = (2 * 37824 + 2304) * 1024


_ORACLE = {}      # inputs of the CPU baseline, inherited by the forked workers


def _oracle_spec(cfg):
    from oracle import ddmpc_oracle as orc
    return orc.QPSpec(n=cfg["n"], m=cfg["m"], p=cfg["p"], L=cfg["L"], Q=cfg["Q"] * np.eye(cfg["p"] * cfg["L"]),
                      R=cfg["R"] * np.eye(cfg["m"] * cfg["L"]), u_s=cfg["u_s"], y_s=cfg["y_s"], robust=cfg["robust"],
                      eps_max=cfg["eps_max"], lamb_alpha=cfg["lamb_alpha"], lamb_sigma=cfg["lamb_sigma"], c=cfg["c"],
                      slack=cfg["slack"], tec=cfg["tec"])


def _oracle_chunk(bounds):
    """Worker: full-space oracle solves of instances [lo, hi), one BLAS thread."""
    from oracle import ddmpc_oracle as orc
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(limits=1)
    except Exception:       # pragma: no cover
        pass
    lo, hi = bounds
    g = _ORACLE
    u = np.empty((hi - lo, g["spec"].L * g["spec"].m)); c = np.empty(hi - lo)
    for i, b in enumerate(range(lo, hi)):
        sol = orc.solve_fullspace(g["spec"], g["u_d"][b], g["y_d"][b], g["up"][b], g["yp"][b])
        u[i] = sol.optimal_u; c[i] = sol.cost
    return u, c


def cpu_baseline(cfg, u_d, y_d, up, yp, n_sample):
    """Time the CPU oracle (full-space KKT restatement of the reference QP) on a bounded sample of the same
    workload, instances spread over `cores` single-threaded worker processes (forked BEFORE the GPU runtime is
    initialised).  Returns the baseline record and the oracle's (optimal_u, cost) for the parity check."""
    import multiprocessing as mp
    from oracle import ddmpc_oracle as orc
    from oracle import reduced_form as rf
    try:
        from threadpoolctl import threadpool_limits
    except Exception:       # pragma: no cover
        threadpool_limits = None
    cores = min(16, os.cpu_count() or 1)
    spec = _oracle_spec(cfg)
    _ORACLE.update(spec=spec, u_d=u_d, y_d=y_d, up=up, yp=yp)
    edges = np.linspace(0, n_sample, 4 * cores + 1).astype(int)
    chunks = [(int(edges[i]), int(edges[i + 1])) for i in range(4 * cores) if edges[i + 1] > edges[i]]
    with mp.get_context("fork").Pool(cores) as pool:
        pool.map(_oracle_chunk, [(0, 1)] * cores)              # start the workers outside the timed region
        t0 = time.perf_counter()
        parts = pool.map(_oracle_chunk, chunks, chunksize=1)
        run_t = time.perf_counter() - t0
    u_ref = np.concatenate([p[0] for p in parts]); c_ref = np.concatenate([p[1] for p in parts])

    def timed(fn, n):
        t0 = time.perf_counter()
        for b in range(n):
            fn(spec, u_d[b], y_d[b], up[b], yp[b])
        return time.perf_counter() - t0

    # the same oracle on ONE thread of one process (SURVEY 8d asks for both), and -- for context only -- the same
    # CPU on the REDUCED r x r formulation the GPU kernels use (oracle/reduced_form.py: numpy BLAS Gram + LAPACK
    # Cholesky): the algorithmic change alone, without the GPU
    n_one, n_red = min(n_sample, 128), min(n_sample, 256)
    if threadpool_limits is not None:
        with threadpool_limits(limits=1):
            one_t = timed(orc.solve_fullspace, n_one)
            red_t = timed(rf.solve_reduced, n_red)
    else:
        one_t = timed(orc.solve_fullspace, n_one)
        red_t = timed(rf.solve_reduced, n_red)
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    rec = dict(value=n_sample / run_t, unit="QP solves/s", cores=cores, kind="port", cpu_model=cpu_model,
               host_cpus=os.cpu_count(),
               sample="%d cold solves (first %d instances of the batch), full-space dense KKT in numpy/LAPACK, %d worker "
                      "processes x 1 thread, %.1f s" % (n_sample, n_sample, cores, run_t),
               single_thread_value=n_one / one_t,
               reduced_form_single_thread_value=n_red / red_t,
               reduced_form_note="same CPU, one thread, the reduced r x r formulation of oracle/reduced_form.py (numpy BLAS "
                                 "Gram + LAPACK Cholesky) on %d instances: context for the algorithmic share of the speed-up" % n_red)
    return rec, u_ref, c_ref


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch-per-gpu", type=int, default=4096)
    ap.add_argument("--slack", choices=["none", "convex"], default="none")
    ap.add_argument("--cpu-sample", type=int, default=4096)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-warm", action="store_true", help="skip the secondary warm-step measurement")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="development only: run all ranks on device 0 with the gloo backend (exercises the "
                         "multi-rank code path on a one-GPU box; numbers are meaningless)")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    from direct_data_driven_mpc_amd import _lib as L
    from direct_data_driven_mpc_amd.distributed import gather_results, shard_bounds
    from direct_data_driven_mpc_amd.engine import BatchedDDMPC
    from direct_data_driven_mpc_amd.harness import controller_params, generate_batch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (a.gpus, a.gpus))
    # synthetic inputs (host) and, at N=1, the CPU baseline -- before anything touches the GPU runtime, so
    # that the baseline's worker processes can simply be forked
    cfg = controller_params(dict(slack_var_constraint_type=1 if a.slack == "convex" else 0))
    total = a.batch_per_gpu * world
    lo, hi = shard_bounds(total, rank, world)
    B = hi - lo
    data = generate_batch(range(lo, hi), N=cfg["N"])           # seeds = global instance ids
    u_d_h, y_d_h = data["u_d"], data["y_d"]
    n, m, p = cfg["n"], cfg["m"], cfg["p"]
    up_h = u_d_h[:, -n:, :].reshape(B, -1).copy()
    yp_h = y_d_h[:, -n:, :].reshape(B, -1).copy()
    cpu = None
    if world == 1 and rank == 0 and not a.no_cpu_baseline:
        cpu = cpu_baseline(cfg, u_d_h, y_d_h, up_h, yp_h, min(a.cpu_sample, B))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the engine has no CPU fallback")
    if a.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if a.rehearse_on_one_gpu:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)

    eng = BatchedDDMPC(n=n, m=m, p=p, L_=cfg["L"], N=cfg["N"], Q=cfg["Q"], R=cfg["R"], u_s=cfg["u_s"], y_s=cfg["y_s"],
                       batch=B, controller_type=L.ROBUST, slack_type=L.SLACK_CONVEX if a.slack == "convex" else L.SLACK_NONE,
                       eps_max=cfg["eps_max"], lamb_alpha=cfg["lamb_alpha"], lamb_sigma=cfg["lamb_sigma"], c=cfg["c"],
                       use_terminal_constraint=cfg["tec"], device=local_rank)
    u_d = torch.from_numpy(u_d_h).to(dev); y_d = torch.from_numpy(y_d_h).to(dev)
    up = torch.from_numpy(up_h).to(dev); yp = torch.from_numpy(yp_h).to(dev)
    u_opt = torch.empty((B, cfg["L"] * m), dtype=torch.float64, device=dev)
    cost = torch.empty((B,), dtype=torch.float64, device=dev)
    status = torch.empty((B,), dtype=torch.int32, device=dev)
    iters = torch.empty((B,), dtype=torch.int32, device=dev)
    eng.set_data(u_d, y_d)

    def barrier():
        if world > 1:
            if a.rehearse_on_one_gpu:
                dist.barrier()
            else:
                dist.barrier(device_ids=[local_rank])

    for _ in range(a.warmup):
        eng.solve(up, yp, u_opt, cost, status, iters)
    if world > 1 and a.warmup > 0:              # the gather's first call sets up RCCL channels: part of the warm-up
        if a.rehearse_on_one_gpu:
            gather_results(u_opt.cpu(), cost.cpu(), status.cpu(), total)
        else:
            gather_results(u_opt, cost, status, total)
    torch.cuda.synchronize()
    # ---- timed region: exactly K steps + the final gather ------------------------
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(a.steps):
        ev[k][0].record()                       # same stream the kernel is launched on
        eng.solve(up, yp, u_opt, cost, status, iters)
        ev[k][1].record()
    if world > 1:
        if a.rehearse_on_one_gpu:
            g_u, g_c, g_s = gather_results(u_opt.cpu(), cost.cpu(), status.cpu(), total)
        else:
            g_u, g_c, g_s = gather_results(u_opt, cost, status, total)
    torch.cuda.synchronize(); barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=("cpu" if a.rehearse_on_one_gpu else dev))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    kern_ms = float(np.mean([s.elapsed_time(e) for s, e in ev]))
    st = status.cpu().numpy()
    n_bad = int(np.count_nonzero(st != 0))
    if world > 1:
        n_bad = int(np.count_nonzero(g_s.cpu().numpy() != 0))
        assert g_u.shape[0] == total and torch.equal(g_u[lo:hi].cpu(), u_opt.cpu()), "gather mismatch"

    # ---- secondary metric (SURVEY 8d "warm" step), outside the timed region: rank 0, one GPU's share.
    # After ddmpc_prepare a control step only evaluates the per-instance affine law (slack NONE).
    warm = None
    if rank == 0 and world == 1 and a.slack == "none" and not a.no_warm:      # N=1 only: the other ranks must not wait on it
        u_cold = u_opt.clone(); c_cold = cost.clone()
        torch.cuda.synchronize(); tp = time.perf_counter()
        eng.prepare()
        torch.cuda.synchronize(); prep_ms = (time.perf_counter() - tp) * 1e3
        for _ in range(5):
            eng.step(up, yp, u_opt, cost, status, iters)
        kw = 100
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(kw):
            eng.step(up, yp, u_opt, cost, status, iters)
        e1.record(); torch.cuda.synchronize()
        wms = e0.elapsed_time(e1) / kw
        nf, r = n * (m + p), (m + p) * (cfg["L"] + n)
        wbytes = 8.0 * ((nf + 1) * r + nf + cfg["L"] * m + 1) + 8.0       # gain + past window in, u_opt/cost/status/iters out
        gbps = wbytes * B / (wms * 1e-3) / 1e9
        warm = {"value": B / (wms * 1e-3), "unit": "control steps/s per GPU", "ms_per_step": wms, "prepare_ms": prep_ms,
                "kernel": "ddmpc_warm_step_kernel", "bytes_per_step": wbytes,
                "roofline": {"bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                             "frac": gbps / HBM_PEAK_GBPS,
                             "traffic": (WARM_PMC_TRAFFIC_BYTES_DEFAULT if a.batch_per_gpu == 4096 else None)},
                "max_rel_diff_vs_cold_u": float((u_opt - u_cold).abs().max() / u_cold.abs().max()),
                "max_rel_diff_vs_cold_cost": float(((cost - c_cold).abs() / c_cold.abs()).max())}
        u_opt.copy_(u_cold); cost.copy_(c_cold)

    if rank == 0:
        flops, bytes_ = eng.cost_model()
        achieved = flops * B / (kern_ms * 1e-3) / 1e12
        out = {
            "metric": "QP solves/sec (batched), four-tank robust DD-MPC L=30 N=400",
            "value": total * a.steps / dt, "unit": "QP solves/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "four-tank robust DD-MPC cold solve, L=30 N=400 n=4 m=p=2, slack %s, TEC, "
                                   "batch=%d noise seeds per GPU (BASELINE configs[1])" % (a.slack.upper(), a.batch_per_gpu),
                       "global_batch": total, "parallelism": "instances sharded dp%d, no data-path collective, one final all-gather" % world,
                       "kernel": eng.kernel_name(), "non_optimal_instances": n_bad},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_MFMA_PEAK_TFLOPS,
                         "traffic": (PMC_TRAFFIC_BYTES_DEFAULT if (a.batch_per_gpu == 4096 and a.slack == "none") else None),
                         "kernel_ms": kern_ms, "flops_per_solve": flops, "hbm_bytes_per_solve": bytes_,
                         "hbm_GBps_algorithmic": bytes_ * B / (kern_ms * 1e-3) / 1e9},
        }
        if warm is not None:
            out["warm_step"] = warm
        if cpu is not None:
            base, u_ref, c_ref = cpu
            ns = u_ref.shape[0]
            u_gpu, c_gpu = u_opt.cpu().numpy()[:ns], cost.cpu().numpy()[:ns]
            eu = float(np.max(np.max(np.abs(u_gpu - u_ref), axis=1) / np.max(np.abs(u_ref), axis=1)))
            ec = float(np.max(np.abs(c_gpu - c_ref) / np.abs(c_ref)))
            out["cpu_baseline"] = base
            out["parity"] = dict(max_rel_err_u=eu, max_rel_err_cost=ec, checked=ns, tol_u=1e-8, tol_cost=1e-9)
        print(json.dumps(out), flush=True)
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
