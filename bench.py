#!/usr/bin/env python3
"""bench.py -- MAPF-GPT per-step hot path on MI355X: env step + observation tokenizer + GPT forward + sample.

    python bench.py --gpus 1 --steps 16 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path over every instance of the workload (one env step of every agent).
Arithmetic: --precision f16x3 (default; split-fp16 3-pass MFMA with fp32 accumulate, logits within 1e-5 of the
reference fp32 forward -- tests/test_gpu_gpt.py), f32 (exact fp32 MFMA) or bf16 (reduced precision, not the headline).
Workload (BASELINE.json configs[1]): map validation-mazes-seed-000, 64 agents, MAPF-GPT-2M shape,
256 parallel instances PER GPU (weak scaling: instances shard across ranks with no per-step
collective; one metrics all_gather after the timed region).  Synthetic data: seeded starts/goals
(instance i = seed i) and seeded random-init weights of the 2M architecture (released checkpoints
need network).  Inputs are resident in HBM before the timed region; nothing crosses PCIe inside it.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

WORKLOADS = {
    # name: (map, agents, instances per GPU, model, max_episode_steps)   -- BASELINE.json configs
    "cfg1": ("validation-random-seed-000", 32, 1, "2M", 128),
    "cfg2": ("validation-mazes-seed-000", 64, 256, "2M", 128),
    "cfg3": ("wfi_warehouse", 192, 64, "6M", 128),
    # cfg4 = BASELINE configs[3]: 4096 instances over 8 GPUs = 512 per GPU; every instance has its own synthetic map, half
    # Bernoulli "random" (obstacle density U[0.1, 0.3]) and half "maze" (wall density ~0.3), 40 x 40 cells so that 128 agents fit
    # (the reference's 17-21-cell eval maps hold at most 64)
    "cfg4": ("synthetic-random+maze-40x40", 128, 512, "6M", 128),
    "cfg5": ("Berlin_1_256_00", 256, 128, "85M", 256),
}
PEAK_TFLOPS = {"f32": 157.3, "f16x3": 2500.0, "bf16": 2500.0}   # MI355X_MICROARCH.md: dense MFMA peaks
PEAK_HBM_GBS = 8000.0
TOKENIZER_BYTES_PER_ROW = 694        # 242 window + 14 own + 182 neighbours + 256 row (uint8 tokens)


def flops_per_row(model_args):
    """Algorithmic (reference-executed) flops per agent-step and per kernel class."""
    L, C, T, V = model_args["n_layer"], model_args["n_embd"], 256, 67
    per_layer = {"gpt_gemm_qkv": 6 * C * C * T, "gpt_attention": 4 * T * T * C, "gpt_gemm_attn_proj": 2 * C * C * T,
                 "gpt_gemm_mlp_fc": 8 * C * C * T, "gpt_gemm_mlp_proj": 8 * C * C * T, "gpt_mlp_fused": 16 * C * C * T,
                 "gpt_ln_qkv_fused": 6 * C * C * T}
    total = L * (24 * C * C * T + 4 * T * T * C) + 2 * C * V
    return total, per_layer


def cpu_baseline(map_name, n_agents, model, budget_s=12.0):
    """Oracle (C env + tokenizer restatement, PyTorch-CPU fp32 forward = the ops the reference executes) timed on
    the host cores, bounded sample of the same workload."""
    from mapf_gpt_amd import maps, weights
    from mapf_gpt_amd.runner import make_instances
    from oracle import gpt_oracle
    from oracle import oracle as orc
    n_inst = 2
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    grid, s_ok, g_ok = maps.load_named(map_name)
    pos, goal = make_instances(grid, n_inst, n_agents, 0, s_ok, g_ok)
    args = weights.model_args(model)
    sd = gpt_oracle.to_torch(weights.synthetic_state_dict(model, seed=0))
    gens = [orc.OracleGenerator(grid) for _ in range(n_inst)]
    p, g = pos.numpy().astype(np.int32).copy(), goal.numpy().astype(np.int32)
    last = np.full((n_inst, n_agents), -1, np.int32)
    for i in range(n_inst):
        gens[i].create_agents(p[i], g[i])
    # PyTorch's default (one thread per host cpu) collapses on a 2-socket 256-thread box; give the CPU path its
    # best shot: try a few intra-op thread counts on one forward each and keep the fastest.
    probe_rows = np.concatenate([gens[i].generate_observations() for i in range(n_inst)])
    best_t, best_n = None, 1
    for nt in sorted({min(ncpu, c) for c in (8, 16, 32, 64)}):
        torch.set_num_threads(nt)
        with torch.no_grad():
            gpt_oracle.forward_logits(sd, args, probe_rows[:16])
            t0 = time.perf_counter()
            gpt_oracle.forward_logits(sd, args, probe_rows)
            dt = time.perf_counter() - t0
        if best_t is None or dt < best_t:
            best_t, best_n = dt, nt
    torch.set_num_threads(best_n)
    steps, t_tok, t_fwd, t_env = 0, 0.0, 0.0, 0.0
    t_start = time.perf_counter()
    while True:
        t0 = time.perf_counter()
        rows = []
        for i in range(n_inst):
            gens[i].update_agents(p[i], g[i], last[i])
            rows.append(gens[i].generate_observations())
        t1 = time.perf_counter()
        with torch.no_grad():
            logits = gpt_oracle.forward_logits(sd, args, np.concatenate(rows))
            act = torch.multinomial(gpt_oracle.act_probs(logits), 1).squeeze(1).numpy().astype(np.int32).reshape(n_inst, n_agents)
        t2 = time.perf_counter()
        for i in range(n_inst):
            p[i], _ = orc.env_step(grid, p[i], g[i], act[i])
        t3 = time.perf_counter()
        last = act
        if steps > 0:            # first step = warmup (allocator, thread pool)
            t_tok += t1 - t0; t_fwd += t2 - t1; t_env += t3 - t2
        steps += 1
        if steps >= 3 and time.perf_counter() - t_start > budget_s:
            break
    timed = steps - 1
    total = t_tok + t_fwd + t_env
    cores = torch.get_num_threads()   # read BEFORE touching the reference tokenizer: its ctor calls omp_set_num_threads(1) (h:115)
    ref_tok = None
    ref_dir = os.path.join(ROOT, "oracle", "_ref")
    try:                          # the REAL reference tokenizer, if its build travelled (oracle/_ref)
        sys.path.insert(0, ref_dir)
        import observation_generator as og
        gen = og.ObservationGenerator(grid.astype(int).tolist(), og.InputParameters(20, 13, 5, 256, 5, 5, 64, False))
        pl, gl = [tuple(map(int, x)) for x in p[0]], [tuple(map(int, x)) for x in g[0]]
        gen.create_agents(pl, gl)
        t0 = time.perf_counter()
        for _ in range(20):
            gen.update_agents(pl, gl, [0] * n_agents)
            gen.generate_observations()
        ref_tok = (time.perf_counter() - t0) / 20 / n_agents * 1e6
    except Exception:
        pass
    return {"value": n_inst * n_agents * timed / total, "unit": "agent-steps/s", "cores": cores,
            "kind": "port",
            "sample": f"{n_inst} instances x {n_agents} agents x {timed} steps of the same workload, {model} fp32 PyTorch-CPU forward "
                      f"+ C oracle env/tokenizer; best of 8/16/32/64 torch threads on {ncpu} host cpus",
            "split_ms_per_step": {"tokenizer": 1e3 * t_tok / timed, "forward+sample": 1e3 * t_fwd / timed, "env": 1e3 * t_env / timed},
            "reference_tokenizer_us_per_agent": ref_tok}


def tokenizer_large_launch(grid, s_ok, g_ok, n_agents, local_rank, target_rows=524288, reps=20):
    """the tokenizer's HBM roofline is to be read on launches of >= 1e5 rows (cfg4/5-sized per-GPU shards), not on
    cfg2's 16 384 rows.  Same map and agent count as the workload, instances replicated up to ~5e5 rows, HIP-event timing
    of the tokens kernel over `reps` launches (inputs resident in HBM)."""
    import torch
    from mapf_gpt_amd import _lib
    from mapf_gpt_amd.observation_generator import BatchedTokenizer
    from mapf_gpt_amd.runner import make_instances
    dev = f"cuda:{local_rank}"
    n_inst = max(1, target_rows // n_agents)
    base = min(n_inst, 256)
    pos, goal = make_instances(grid, base, n_agents, 0, s_ok, g_ok)
    k = (n_inst + base - 1) // base
    pos = pos.repeat(k, 1, 1)[:n_inst].contiguous().to(dev)
    goal = goal.repeat(k, 1, 1)[:n_inst].contiguous().to(dev)
    tok = BatchedTokenizer(grid, n_inst, n_agents, device=dev)
    tok.create_agents(pos, goal)
    act = torch.zeros((n_inst, n_agents), dtype=torch.int32, device=dev)
    out = torch.empty((n_inst * n_agents, 256), dtype=torch.uint8, device=dev)
    for _ in range(3):
        tok.update_agents(pos, goal, act, goals_may_change=False)
        tok.generate_observations(out)
    _lib.prof_reset()
    _lib.prof_enable(True)
    for _ in range(reps):
        tok.update_agents(pos, goal, act, goals_may_change=False)
        tok.generate_observations(out)
    _lib.prof_enable(False)
    p = _lib.prof_read()
    rows = n_inst * n_agents
    ms, n = p["tok_generate_observations"]
    ach = TOKENIZER_BYTES_PER_ROW * rows / (ms / n * 1e-3) / 1e9
    del tok, out
    torch.cuda.empty_cache()
    traffic = None                  # HBM bytes per launch from the committed PMC passes of exactly this launch shape
    tf = os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")
    if os.path.exists(tf):
        t = json.load(open(tf)).get(f"tok_generate_observations_{rows}_rows")
        if t:
            traffic = {"hbm_bytes_per_launch": t["fetch_corrected_x2"] + t["write"], "fetch_raw": t["fetch_raw"], "write": t["write"],
                       "source": "profiles/r01_hbm_traffic.json (rocprofv3 PMC, FETCH_SIZE x2 gfx950 correction)"}
    return {"kernel": "tok_generate_observations", "bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": ach / PEAK_HBM_GBS, "traffic": traffic, "avg_launch_ms": ms / n, "launches": n, "rows_per_launch": rows,
            "algorithmic_bytes_per_row": TOKENIZER_BYTES_PER_ROW,
            "note": "694 B/row = SURVEY 8d (u16 window 242 + own record 14 + 13 neighbour records 182 + uint8 row 256); the kernel "
                    "itself reads one-byte fields when every distance fits (573 B/row by its own layout)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--precision", default=os.environ.get("MGPT_BENCH_PRECISION", "f16x3"), choices=["f32", "f16x3", "bf16"])
    ap.add_argument("--instances", type=int, default=0, help="instances per GPU (default: the workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-tokenizer-leg", action="store_true", help="skip the >=1e5-row tokenizer roofline launch")
    ap.add_argument("--no-prof", action="store_true", help="skip the per-kernel HIP-event hooks")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # MGPT_BENCH_BACKEND=gloo + MGPT_BENCH_SHARE_GPU=1: control-flow test of the N>1 path on a one-GPU box
    # (all ranks on cuda:0, collectives on CPU tensors); the real runs use RCCL ("nccl") with one GPU per rank.
    backend = os.environ.get("MGPT_BENCH_BACKEND", "nccl")
    if os.environ.get("MGPT_BENCH_SHARE_GPU"):
        local_rank = 0
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    coll_dev = "cuda" if backend == "nccl" else "cpu"
    assert a.gpus == world, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local_rank)

    from mapf_gpt_amd import _lib, maps, weights
    from mapf_gpt_amd.model import build_model
    from mapf_gpt_amd.runner import BatchedRunner, gather_metrics, make_instances, shard_range

    map_name, n_agents, inst_per_gpu, model, max_steps = WORKLOADS[a.workload]
    if a.instances:
        inst_per_gpu = a.instances
    n_total = inst_per_gpu * world
    lo, hi = shard_range(n_total, rank, world)
    rows = (hi - lo) * n_agents
    chunk = min(rows, 4096 if model != "85M" else 1024)
    net = build_model(model, seed=0, max_rows=chunk, precision=a.precision, device=f"cuda:{local_rank}")
    if a.workload == "cfg4":                     # one map per instance, seeded by the global instance id
        import numpy as np
        gl, pl, gll = [], [], []
        for i in range(lo, hi):
            rng = np.random.Generator(np.random.PCG64([i, 4]))
            obst = maps.random_map(40, 40, float(rng.uniform(0.1, 0.3)), 1000 + i) if i % 2 == 0 else maps.maze_map(40, 40, 1000 + i)
            g = maps.pad(obst)
            p_, g_ = maps.place_agents(g, n_agents, i)
            gl.append(g); pl.append(p_); gll.append(g_)
        grid, s_ok, g_ok = np.stack(gl), None, None
        pos, goal = torch.from_numpy(np.stack(pl)), torch.from_numpy(np.stack(gll))
    else:
        grid, s_ok, g_ok = maps.load_named(map_name)
        pos, goal = make_instances(grid, hi - lo, n_agents, first_seed=lo, start_ok=s_ok, goal_ok=g_ok)
    run = BatchedRunner(grid, hi - lo, n_agents, net, max_episode_steps=max_steps, seed=0, do_sample=True,
                        precision=a.precision, device=f"cuda:{local_rank}", row_offset=lo * n_agents)
    run.reset(pos, goal)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def hot_steps(k):
        for _ in range(k):
            if run.t >= max_steps:               # episode over: new episode (part of the job, stays inside the timing)
                run.reset(pos, goal)
            run.step()

    hot_steps(a.warmup)
    use_prof = not a.no_prof
    if use_prof:
        _lib.prof_reset()
        _lib.prof_enable(True)
    barrier()
    t0 = time.perf_counter()
    hot_steps(a.steps)
    barrier()
    dt = time.perf_counter() - t0
    prof = {}
    if use_prof:
        _lib.prof_enable(False)
        prof = _lib.prof_read()
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    metrics = gather_metrics(run.metrics().to(coll_dev), n_total, rank, world)       # the job's one collective
    torch.cuda.synchronize()

    if rank == 0:
        margs = weights.model_args(model)
        f_total, f_class = flops_per_row(margs)
        value = n_total * n_agents * a.steps / dt
        out = {"metric": "agent-steps/s (env+obs+GPT fwd)", "value": value, "unit": "agent-steps/s", "n_gpus": world,
               "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": a.precision, "data": "synthetic",
               "config": {"workload": f"{a.workload}: {map_name}, {n_agents} agents, MAPF-GPT-{model} shape, "
                                      f"{inst_per_gpu} instances/GPU ({n_total} total), {n_total * n_agents} rows/step",
                          "parallelism": f"instances sharded x{world}, no per-step collective",
                          "gflop_per_agent_step": f_total / 1e9,
                          "mean_ISR_after_run": float(metrics[:, 1].mean().item())}}
        if prof:
            # algorithmic (reference-executed) flops per step of every kernel class that actually ran; fused classes
            # carry the flops of everything they absorbed (LN+QKV and the out-projection live in "gpt_attention" when
            # their own classes are absent).  The last-layer shortcut is OUR saving: flops stay the reference's.
            L_ = margs["n_layer"]
            cls = {k: f_class[k] * L_ * rows for k in f_class if k in prof}
            if "gpt_attention" in prof:
                if "gpt_gemm_qkv" not in prof and "gpt_ln_qkv_fused" not in prof:
                    cls["gpt_attention"] += f_class["gpt_gemm_qkv"] * L_ * rows
                if "gpt_gemm_attn_proj" not in prof:
                    cls["gpt_attention"] += f_class["gpt_gemm_attn_proj"] * L_ * rows
            if cls:
                dom = max(cls, key=lambda k: prof[k][0])
                ms, n = prof[dom]
                ach = cls[dom] * a.steps / (ms * 1e-3) / 1e12
                traffic = None      # HBM bytes per launch from the committed PMC passes (same workload), if this is that workload
                tf = os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")
                if a.workload == "cfg2" and a.precision == "f16x3" and os.path.exists(tf):
                    t = json.load(open(tf)).get(dom)
                    if t:
                        traffic = {"hbm_bytes_per_launch": t["fetch_corrected_x2"] + t["write"], "fetch_raw": t["fetch_raw"],
                                   "write": t["write"], "source": "profiles/r01_hbm_traffic.json (rocprofv3 PMC, FETCH_SIZE x2 gfx950 correction)"}
                out["roofline"] = {"kernel": dom, "bound": "mfma", "achieved": ach, "peak": PEAK_TFLOPS[a.precision],
                                   "unit": "TFLOP/s", "frac": ach / PEAK_TFLOPS[a.precision], "traffic": traffic,
                                   "avg_launch_ms": ms / n, "launches": n,
                                   "algorithmic_gflop_per_launch": cls[dom] * a.steps / n / 1e9,
                                   "mfma_issue_frac": (3.0 if a.precision == "f16x3" else 1.0) * ach / PEAK_TFLOPS[a.precision],
                                   "note": "algorithmic flops (reference-executed) / HIP-event time of the class over the timed region"
                                           + ("; f16x3 issues 3 fp16 MFMAs per product against the fp16 dense peak: frac counts the reference's flops once, mfma_issue_frac counts the issued ones" if a.precision == "f16x3" else "")}
            if "tok_generate_observations" in prof:
                ms, n = prof["tok_generate_observations"]
                ach = TOKENIZER_BYTES_PER_ROW * rows / (ms / n * 1e-3) / 1e9
                out["roofline_tokenizer"] = {"kernel": "tok_generate_observations", "bound": "hbm", "achieved": ach,
                                             "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ach / PEAK_HBM_GBS, "traffic": None,
                                             "avg_launch_ms": ms / n, "launches": n, "rows_per_launch": rows,
                                             "note": "the workload's own launch (latency-bound when rows_per_launch < 1e5)"}
            if world == 1 and not a.no_tokenizer_leg and a.workload != "cfg4":
                out["roofline_tokenizer_large"] = tokenizer_large_launch(grid, s_ok, g_ok, n_agents, local_rank)
            out["kernel_ms_per_step"] = {k: v[0] / a.steps for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])}
        if world == 1 and not a.no_cpu_baseline and a.workload != "cfg4":
            out["cpu_baseline"] = cpu_baseline(map_name, n_agents, model)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
