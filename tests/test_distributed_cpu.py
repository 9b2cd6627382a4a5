"""world_size-2 gloo test (CPU) of the multi-GPU path: shard -> solve locally ->
one all-gather; gathered result must equal the single-process result bit for bit.
The local solve is stood in by the CPU oracle (there is no GPU here)."""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

TOTAL = 5      # odd on purpose: ragged shards


def _solve_block(lo, hi):
    from oracle import ddmpc_oracle as orc
    spec = orc.spec_from_params(L=10, N=120)
    us, cs, ss = [], [], []
    for seed in range(lo, hi):
        inst = orc.generate_instance(seed, N=120)
        u_d, y_d = inst["u_d"], inst["y_d"]
        sol = orc.solve_fullspace(spec, u_d, y_d, u_d[-4:].reshape(-1), y_d[-4:].reshape(-1))
        us.append(sol.optimal_u); cs.append(sol.cost); ss.append(0)
    return (torch.tensor(np.array(us)).reshape(hi - lo, -1), torch.tensor(np.array(cs)).reshape(hi - lo),
            torch.tensor(np.array(ss, dtype=np.int32)).reshape(hi - lo))


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from direct_data_driven_mpc_amd.distributed import gather_results, shard_bounds
    lo, hi = shard_bounds(TOTAL, rank, world)
    u, c, s = _solve_block(lo, hi)
    gu, gc, gs = gather_results(u, c, s, TOTAL)
    if rank == 0:
        q.put((gu.numpy(), gc.numpy(), gs.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_gather_matches_single_process():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    gu, gc, gs = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    u, c, s = _solve_block(0, TOTAL)
    assert gu.shape == (TOTAL, 20)
    assert np.array_equal(gu, u.numpy()) and np.array_equal(gc, c.numpy()) and np.array_equal(gs, s.numpy())


def test_bench_batch_defaults_name_the_baseline_configs():
    # bench.py: --gpus 1 is BASELINE configs[1] (4096 on one GPU); --gpus N > 1 without --batch-per-gpu is configs[2]'s
    # sharding, 32,768 per GPU = 262,144 over 8 -- what the driver's N = 2, 4, 8 runs measure
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.default_batch_per_gpu(1) == 4096
    assert [mod.default_batch_per_gpu(g) * g for g in (2, 4, 8)] == [65536, 131072, 262144]
    from direct_data_driven_mpc_amd.distributed import shard_bounds
    assert shard_bounds(262144, 5, 8) == (5 * 32768, 6 * 32768)
