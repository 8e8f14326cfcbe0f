"""N>1 path on CPU: two gloo ranks shard instances, simulate their shard with the ORACLE (test
infrastructure), and exchange per-instance metrics with the job's single collective."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _episode_metrics(grid, pos, goal, steps, seed):
    """Oracle-driven random-policy episode -> [CSR, ISR, SoC, makespan, ep_length] (same definitions as env.hip)."""
    from oracle import oracle as orc
    rng = np.random.Generator(np.random.PCG64(seed))
    p, g = pos.astype(np.int32), goal.astype(np.int32)
    n = len(p)
    arrive = np.where((p == g).all(1), 0, -1)
    t = 0
    for t in range(1, steps + 1):
        was = (p == g).all(1)
        p, k = orc.env_step(grid, p, g, rng.integers(0, 5, n))
        on = (p == g).all(1)
        arrive = np.where(on & ~was, t, np.where(on, arrive, -1))
        if k == n:
            break
    on = (p == g).all(1)
    ta = np.where(on, np.maximum(arrive, 0), t)
    return np.array([float(on.all()), on.mean(), ta.sum(), ta.max(), t], dtype=np.float32)


def _worker(rank, world, port, n_total, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mapf_gpt_amd import maps
    from mapf_gpt_amd.runner import gather_metrics, make_instances, shard_range
    grid, s_ok, g_ok = maps.load_named("validation-random-seed-000")
    lo, hi = shard_range(n_total, rank, world)
    pos, goal = make_instances(grid, hi - lo, 8, first_seed=lo, start_ok=s_ok, goal_ok=g_ok)
    local = torch.from_numpy(np.stack([_episode_metrics(grid, pos[i].numpy(), goal[i].numpy(), 16, seed=lo + i)
                                       for i in range(hi - lo)]))
    full = gather_metrics(local, n_total, rank, world)
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), full.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_metrics_gather(tmp_path):
    n_total, world = 5, 2            # uneven split: 3 + 2
    mp.start_processes(_worker, args=(world, _free_port(), n_total, str(tmp_path)), nprocs=world, join=True,
                       start_method="spawn")
    r0, r1 = np.load(tmp_path / "rank0.npy"), np.load(tmp_path / "rank1.npy")
    assert r0.shape == (n_total, 5) and np.array_equal(r0, r1)
    sys.path.insert(0, ROOT)
    from mapf_gpt_amd import maps
    from mapf_gpt_amd.runner import make_instances
    grid, s_ok, g_ok = maps.load_named("validation-random-seed-000")
    pos, goal = make_instances(grid, n_total, 8, 0, s_ok, g_ok)
    single = np.stack([_episode_metrics(grid, pos[i].numpy(), goal[i].numpy(), 16, seed=i) for i in range(n_total)])
    assert np.array_equal(r0, single)          # sharded run == single-process run, instance by instance
    assert (r0[:, 4] <= 16).all() and (r0[:, 1] >= 0).all() and (r0[:, 1] <= 1).all()


def _worker_one(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mapf_gpt_amd.runner import gather_metrics
    local = torch.arange(15, dtype=torch.float32).reshape(3, 5)
    assert gather_metrics(local, 3, 0, 1) is local                                # world = 1: no collective unless forced
    np.save(os.path.join(out_dir, "one.npy"), gather_metrics(local, 3, 0, 1, force=True).numpy())
    dist.destroy_process_group()


def test_forced_collective_with_one_rank(tmp_path):
    """gather_metrics(force=True) at world = 1 goes through all_gather of a one-rank group (the path bench.py takes under
    MGPT_BENCH_FORCE_COLLECTIVE to exercise the RCCL branch on a one-GPU box) and returns the records unchanged."""
    mp.start_processes(_worker_one, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True, start_method="spawn")
    assert np.array_equal(np.load(tmp_path / "one.npy"), np.arange(15, dtype=np.float32).reshape(3, 5))
