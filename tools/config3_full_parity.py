"""BASELINE configs[2] at its full size -- 262,144 instances = 8 shards of 32,768 -- solved shard by shard on ONE GPU
(each shard is exactly what rank r of an 8-GPU job owns: direct_data_driven_mpc_amd.distributed.shard_bounds) and
EVERY instance checked against the compiled CPU restatement (oracle/ddmpc_oracle_c.c, all host cores).

    python tools/config3_full_parity.py [--shards 8] [--slack none|convex]

--slack convex: the same instances with the slack box (controller.py:631-677) -- on the GPU the active-set iterations after the
first run as rank-k updates of the first factor; the iteration counts are compared too.
"""
import argparse, os, sys, time
import multiprocessing as mp
import numpy as np
sys.path.insert(0, ".")
import bench                                                   # its QPSpec builder for the benchmark parameters
from oracle import oracle_c                                    # compiled restatement of the reduced form (OpenMP)
from direct_data_driven_mpc_amd.distributed import shard_bounds
from direct_data_driven_mpc_amd.harness import controller_params, generate_batch

ap = argparse.ArgumentParser(); ap.add_argument("--shards", type=int, default=8); ap.add_argument("--per-shard", type=int, default=32768)
ap.add_argument("--slack", default="none", choices=("none", "convex"))
a = ap.parse_args()
cfg = controller_params(dict(slack_var_constraint_type=1)) if a.slack == "convex" else controller_params()
total = a.shards * a.per_shard
n, m, p = cfg["n"], cfg["m"], cfg["p"]
refs, inputs = [], []
t0 = time.perf_counter()
for rank in range(a.shards):                                   # CPU side first (fork before HIP is initialised)
    lo, hi = shard_bounds(total, rank, a.shards)
    d = generate_batch(range(lo, hi), N=cfg["N"])
    up = d["u_d"][:, -n:, :].reshape(hi - lo, -1).copy(); yp = d["y_d"][:, -n:, :].reshape(hi - lo, -1).copy()
    ts = time.perf_counter()
    u_ref, c_ref, st_ref, it_ref = oracle_c.solve_batch(bench._oracle_spec(cfg), cfg["N"], d["u_d"], d["y_d"], up, yp, threads=bench.host_cores())
    assert not np.count_nonzero(st_ref)
    refs.append((u_ref, c_ref, it_ref)); inputs.append((d["u_d"], d["y_d"], up, yp))
    print("shard %d: C restatement %.0f solves/s on %d threads" % (rank, (hi - lo) / (time.perf_counter() - ts), bench.host_cores()), flush=True)
t_cpu = time.perf_counter() - t0

import torch
from direct_data_driven_mpc_amd import _lib as L
from direct_data_driven_mpc_amd.engine import BatchedDDMPC
dev = torch.device("cuda", 0)
worst_u = worst_c = 0.0; bad = 0; t_gpu = 0.0; it_diff = 0; it_hist = {}
for rank in range(a.shards):
    u_d, y_d, up, yp = inputs[rank]
    B = u_d.shape[0]
    eng = BatchedDDMPC(n=n, m=m, p=p, L_=cfg["L"], N=cfg["N"], Q=cfg["Q"], R=cfg["R"], u_s=cfg["u_s"], y_s=cfg["y_s"], batch=B,
                       controller_type=L.ROBUST, slack_type=L.SLACK_CONVEX if a.slack == "convex" else L.SLACK_NONE, eps_max=cfg["eps_max"], lamb_alpha=cfg["lamb_alpha"],
                       lamb_sigma=cfg["lamb_sigma"], c=cfg["c"])
    t = lambda x: torch.from_numpy(x).to(dev)
    tud, tyd, tup, typ = t(u_d), t(y_d), t(up), t(yp)
    eng.set_data(tud, tyd)
    out = eng.solve(tup, typ); torch.cuda.synchronize()
    t1 = time.perf_counter(); out = eng.solve(tup, typ, *out); torch.cuda.synchronize(); t_gpu += time.perf_counter() - t1
    u = out[0].cpu().numpy(); c = out[1].cpu().numpy(); st = out[2].cpu().numpy()
    u_ref, c_ref, it_ref = refs[rank]
    it = out[3].cpu().numpy()
    it_diff += int(np.count_nonzero(it != it_ref))
    for k, v in zip(*np.unique(it, return_counts=True)):
        it_hist[int(k)] = it_hist.get(int(k), 0) + int(v)
    eu = np.max(np.max(np.abs(u - u_ref), axis=1) / np.max(np.abs(u_ref), axis=1)); ec = np.max(np.abs(c - c_ref) / np.abs(c_ref))
    worst_u, worst_c, bad = max(worst_u, eu), max(worst_c, ec), bad + int(np.count_nonzero(st))
    print("shard %d: %d instances, max rel err u %.3e cost %.3e, non-optimal %d" % (rank, B, eu, ec, int(np.count_nonzero(st))), flush=True)
    eng.close()
print("TOTAL %d instances: max rel err u %.3e (tol 1e-8), cost %.3e (tol 1e-9), non-optimal %d; GPU solve time %.1f ms "
      "(%.3e solves/s on one GPU), C restatement %.1f s; slack %s: iteration counts differing from the restatement's %d, histogram %s"
      % (total, worst_u, worst_c, bad, t_gpu * 1e3, total / t_gpu, t_cpu, a.slack, it_diff, sorted(it_hist.items())))
