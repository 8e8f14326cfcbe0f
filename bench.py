#!/usr/bin/env python3
"""bench.py -- MAPF-GPT per-step hot path on MI355X: env step + observation tokenizer + GPT forward + sample.

    python bench.py --gpus 1 --steps 16 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path over every instance of the workload (one env step of every agent).
Arithmetic: --precision f16x3 (default; split-fp16 3-pass MFMA with fp32 accumulate, logits within 1e-5 of the
reference fp32 forward -- tests/test_gpu_gpt.py), f32 (exact fp32 MFMA) or bf16 (reduced precision, not the headline).
Workload: --gpus 1 -> cfg3 = BASELINE.json configs[2] (wfi_warehouse, 192 agents, MAPF-GPT-6M shape -- the model the
north star's target is quoted on --, 64 parallel instances: the largest single-GPU configuration);
--gpus N > 1 -> cfg4 = configs[3]'s per-GPU shard (random+maze mix, 128 agents, 6M, 512 instances per GPU; weak
scaling: instances shard across ranks with no per-step collective; one metrics all_gather after the timed region).
A short cfg2 (configs[1], 2M) run rides along at N = 1 as the secondary key "secondary".
Synthetic data: seeded starts/goals (instance i = seed i) and seeded random-init weights of the named architecture
(released checkpoints need network).  Inputs are resident in HBM before the timed region; nothing crosses PCIe inside it.
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
    # env6M: ONE environment on the 6M model -- how the reference evaluates it (eval_configs/01-random/01-random.yaml:145-148, one
    # environment per worker; inference.py:148-172): 64 agents = 64 rows per step, the small-launch kernels of the 6M shape
    "env6M": ("validation-mazes-seed-000", 64, 1, "6M", 128),
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


def _cpu_worker(job):
    """One CPU-baseline process: its own instance of the workload, `threads` intra-op threads, runs steps until the
    budget is spent.  -> (agent_steps, seconds, split seconds)"""
    map_name, n_agents, model, seed, threads, budget_s, cpus = job
    if cpus and hasattr(os, "sched_setaffinity"):
        os.sched_setaffinity(0, cpus)             # one process per block of cores: no migration, no shared cores
    torch.set_num_threads(threads)
    from mapf_gpt_amd import maps, weights
    from mapf_gpt_amd.runner import make_instances
    from oracle import gpt_oracle
    from oracle import oracle as orc
    grid, s_ok, g_ok = maps.load_named(map_name)
    pos, goal = make_instances(grid, 1, n_agents, seed, s_ok, g_ok)
    args = weights.model_args(model)
    sd = gpt_oracle.to_torch(weights.synthetic_state_dict(model, seed=0))
    gen = orc.OracleGenerator(grid)
    p, g = pos.numpy().astype(np.int32)[0].copy(), goal.numpy().astype(np.int32)[0]
    last = np.full((n_agents,), -1, np.int32)
    gen.create_agents(p, g)
    steps, t_tok, t_fwd, t_env = 0, 0.0, 0.0, 0.0
    t_start = time.perf_counter()
    while True:
        t0 = time.perf_counter()
        gen.update_agents(p, g, last)
        rows = gen.generate_observations()
        t1 = time.perf_counter()
        with torch.no_grad():
            logits = gpt_oracle.forward_logits(sd, args, rows)
            act = torch.multinomial(gpt_oracle.act_probs(logits), 1).squeeze(1).numpy().astype(np.int32)
        t2 = time.perf_counter()
        p, _ = orc.env_step(grid, p, g, act)
        t3 = time.perf_counter()
        last = act
        if steps > 0:            # first step = warmup (allocator, thread pool)
            t_tok += t1 - t0; t_fwd += t2 - t1; t_env += t3 - t2
        steps += 1
        if steps >= 2 and time.perf_counter() - t_start > budget_s:
            break
    timed = steps - 1
    return n_agents * timed, t_tok + t_fwd + t_env, (t_tok, t_fwd, t_env), timed


def usable_cpus():
    """Cores this container may actually burn: the affinity mask capped by the cgroup CPU quota (the round-2 GPU box shows
    256 logical CPUs but cpu.max = 1600000/100000, i.e. 16 cores; a pool sized by the mask was 16x oversubscribed)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return n, quota


def cpu_baseline(map_name, n_agents, model, budget_s=10.0):
    """Oracle (C env + tokenizer restatement, PyTorch-CPU fp32 forward = the ops the reference executes) timed on the host
    cores, bounded sample, two ways: (a) a pool of processes (the reference's own CPU path is a `num_process` pool,
    inference.py:30-31, eval_configs/01-random/01-random.yaml:147-148), 16 intra-op threads each, one instance of the same
    workload per process pinned to its own block of cores; (b) ONE process with up to 32 intra-op threads.  `value` is the
    better of the two.  The core count is the cgroup-quota-capped one (usable_cpus): with 16 usable cores (a) and (b) are the
    same single 16-thread process and only (a) runs."""
    import multiprocessing as mp
    ncpu, quota = usable_cpus()
    threads = min(16, ncpu)
    procs = max(1, ncpu // threads)
    from oracle import oracle as orc
    orc.build()                                   # once, before the pool forks the work out
    allc = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else []
    jobs = [(map_name, n_agents, model, i, threads, budget_s, allc[i * threads:(i + 1) * threads]) for i in range(procs)]
    os.environ["OMP_NUM_THREADS"] = str(threads)  # inherited by the pool: no oversubscribed OpenMP teams at import time
    os.environ["MKL_NUM_THREADS"] = str(threads)
    ctx = mp.get_context("spawn")
    t0 = time.perf_counter()
    with ctx.Pool(procs) as pool:
        res = pool.map(_cpu_worker, jobs)
    wall = time.perf_counter() - t0
    rate_pool = sum(a / t for a, t, _, _ in res)  # processes run concurrently: rates add
    one_threads = min(32, ncpu)
    if procs == 1 and one_threads == threads:
        one = res[0]
    else:
        os.environ["OMP_NUM_THREADS"] = str(one_threads)
        os.environ["MKL_NUM_THREADS"] = str(one_threads)
        with ctx.Pool(1) as pool:
            one = pool.map(_cpu_worker, [(map_name, n_agents, model, 0, one_threads, budget_s, allc[:one_threads])])[0]
    rate_one = one[0] / one[1]
    best_pool = rate_pool >= rate_one
    src = res if best_pool else [one]
    tt = [sum(r[2][k] for r in src) / sum(r[3] for r in src) for k in range(3)]
    ref_tok = None
    ref_dir = os.path.join(ROOT, "oracle", "_ref")
    try:                          # the REAL reference tokenizer, if its build travelled (oracle/_ref)
        from mapf_gpt_amd import maps
        from mapf_gpt_amd.runner import make_instances
        grid, s_ok, g_ok = maps.load_named(map_name)
        pos, goal = make_instances(grid, 1, n_agents, 0, s_ok, g_ok)
        sys.path.insert(0, ref_dir)
        import observation_generator as og
        gen = og.ObservationGenerator(grid.astype(int).tolist(), og.InputParameters(20, 13, 5, 256, 5, 5, 64, False))
        pl, gl = [tuple(map(int, x)) for x in pos[0].tolist()], [tuple(map(int, x)) for x in goal[0].tolist()]
        gen.create_agents(pl, gl)
        t0 = time.perf_counter()
        for _ in range(20):
            gen.update_agents(pl, gl, [0] * n_agents)
            gen.generate_observations()
        ref_tok = (time.perf_counter() - t0) / 20 / n_agents * 1e6
    except Exception:
        pass
    return {"value": max(rate_pool, rate_one), "unit": "agent-steps/s", "cores": procs * threads if best_pool else one_threads,
            "processes": procs if best_pool else 1, "threads_per_process": threads if best_pool else one_threads,
            "usable_cpus": ncpu, "cgroup_cpu_quota": quota, "logical_cpus_visible": os.cpu_count(), "kind": "port",
            "pool_rate": rate_pool, "single_process_rate": rate_one,
            "sample": f"(a) {procs} pinned processes x {threads} threads, each 1 instance x {n_agents} agents of the same workload for ~{budget_s:.0f} s "
                      f"({sum(r[3] for r in res)} timed steps in all, {wall:.0f} s wall incl. start-up): {rate_pool:.0f} agent-steps/s; "
                      f"(b) 1 process x {one_threads} threads, {one[3]} timed steps: {rate_one:.0f} agent-steps/s; "
                      f"{model} fp32 PyTorch-CPU forward + C oracle env/tokenizer",
            "split_ms_per_step_per_process": {"tokenizer": 1e3 * tt[0], "forward+sample": 1e3 * tt[1], "env": 1e3 * tt[2]},
            "reference_tokenizer_us_per_agent": ref_tok}


class ClockPowerSampler:
    """Shader clock and socket power of the GPU while the timed region runs (the energy-bound claim of
    DESIGN section 10 rests on these two numbers, so they ride in the JSON line).  A daemon thread reads the amdgpu hwmon
    files (freq1_input in Hz, power1_average / power1_input in microwatts) every `period` seconds -- microseconds per sample,
    no subprocess; where hwmon is absent it falls back to one `rocm-smi --json` call per 0.5 s."""

    def __init__(self, index=0, period=0.02):
        import glob
        import threading
        self.period, self.samples, self._stop = period, [], threading.Event()
        # every amdgpu card of the box is sampled (a 1-GPU lease still lists the node's other cards in sysfs); the reported one is
        # the card whose PCI address is the HIP device's, else the card that drew the most power during the region
        self.cards = []
        for f in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input")):
            d = os.path.dirname(f)
            pw = next((os.path.join(d, n) for n in ("power1_average", "power1_input") if os.path.exists(os.path.join(d, n))), None)
            bdf = os.path.basename(os.path.realpath(os.path.join(d, "..", "..")))
            self.cards.append((bdf, f, pw))
        self.want_bdf = None
        try:
            pr = torch.cuda.get_device_properties(index)
            self.want_bdf = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        except Exception:
            pass
        self.source = "hwmon" if self.cards else "rocm-smi"
        self.index = index
        self.chosen = None
        self._thread = threading.Thread(target=self._run, daemon=True)

    def _smi(self):
        import subprocess
        try:
            j = json.loads(subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5).stdout)
            c = j.get(f"card{self.index}") or next(iter(j.values()))
            f = p = float("nan")
            for k, v in c.items():
                if "sclk" in k and "(" in str(v):
                    f = float(str(v).split("(")[1].split("Mhz")[0])
                if "Power" in k and "W" in k:
                    p = float(v)
            return f, p
        except Exception:
            return float("nan"), float("nan")

    def _run(self):
        t0 = time.perf_counter()
        while not self._stop.is_set():
            if self.cards:
                row = []
                for _, ff, pf in self.cards:
                    try:
                        f = int(open(ff).read()) / 1e6
                        p = int(open(pf).read()) / 1e6 if pf else float("nan")
                    except Exception:
                        f = p = float("nan")
                    row.append((f, p))
                self.samples.append((time.perf_counter() - t0, row))
                self._stop.wait(self.period)
            else:
                self.samples.append((time.perf_counter() - t0, [self._smi()]))
                self._stop.wait(0.5)

    def __enter__(self):
        self._thread.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        self._thread.join(timeout=10)

    def summary(self, log_path=None):
        if not self.samples:
            return None
        n_cards = len(self.samples[0][1])
        f = np.array([[c[0] for c in row] for _, row in self.samples], dtype=np.float64)      # [samples, cards]
        p = np.array([[c[1] for c in row] for _, row in self.samples], dtype=np.float64)
        pick, how = 0, "only card"
        if n_cards > 1:
            match = [i for i, (bdf, _, _) in enumerate(self.cards) if bdf == self.want_bdf]
            if match:
                pick, how = match[0], f"PCI address {self.want_bdf} of the HIP device"
            else:
                pick, how = int(np.nanargmax(np.nan_to_num(np.nanmean(p, axis=0), nan=-1.0))), "card with the highest mean power in the region"
        self.chosen = pick
        if log_path:
            os.makedirs(os.path.dirname(log_path) or ".", exist_ok=True)
            with open(log_path, "w") as fh:
                fh.write(f"# t_s sclk_MHz socket_power_W -- sampled by bench.py ({self.source}, {how}) during the timed region\n")
                for (t, _), a, b in zip(self.samples, f[:, pick], p[:, pick]):
                    fh.write(f"{t:.3f} {a:.0f} {b:.0f}\n")
        fs, ps = f[:, pick], p[:, pick]
        fs, ps = fs[np.isfinite(fs)], ps[np.isfinite(ps)]
        if len(fs) == 0:
            return None
        out = {"sclk_mhz_mean": float(fs.mean()), "sclk_mhz_min": float(fs.min()), "sclk_mhz_max": float(fs.max()), "samples": int(len(fs)),
               "source": self.source, "card": (self.cards[pick][0] if self.cards else None), "card_chosen_by": how, "cards_sampled": n_cards,
               "period_s": self.period if self.source == "hwmon" else 0.5}
        if len(ps):
            out.update({"socket_power_w_mean": float(ps.mean()), "socket_power_w_max": float(ps.max())})
        return out


LAST_SUFFIX = "_last"      # prof classes of the last layer's launches (token 255 only: model.py:186), timed apart from the full ones


def fold_last(prof):
    """-> (merged, full_only, last_only): the library times the last layer's shortcut launches under `<class>_last`; `merged`
    folds them back into their class (what round 3 reported: the reference's flops over ALL launches of the class)."""
    merged, full, last = {}, {}, {}
    for k, (ms, n) in prof.items():
        if k.endswith(LAST_SUFFIX):
            last[k[:-len(LAST_SUFFIX)]] = (ms, n)
        else:
            full[k] = (ms, n)
    for k in set(full) | set(last):
        a, b = full.get(k, (0.0, 0)), last.get(k, (0.0, 0))
        merged[k] = (a[0] + b[0], a[1] + b[1])
    return merged, full, last


def full_launch_frac(dom, full, f_class_dom, rows_per_launch, peak):
    """The dominant class over its FULL-SIZE launches only: one layer's reference flops for the rows of a launch / the mean
    duration of those launches (the shortcut launches of the last layer, which `frac` averages in while keeping the
    reference's flops, are excluded)."""
    if dom not in full or full[dom][1] == 0:
        return None
    ms, n = full[dom]
    ach = f_class_dom * rows_per_launch / (ms / n * 1e-3) / 1e12
    return {"frac_full_launch": ach / peak, "achieved_full_launch": ach, "avg_full_launch_ms": ms / n, "full_launches": n}


def stream_floor_ms(total_bytes, dev, reps=20):
    """What a launch moving the SAME number of bytes costs with nothing to compute: a device copy of total_bytes / 2 (reads
    half, writes half), torch.cuda events on the current stream.  At cfg4's 45 MB per launch this is what separates "the
    kernel is slow" from "a 20 us launch cannot reach the 8 TB/s plateau" (ramp-up and tail of the grid)."""
    n = int(total_bytes // 2)
    src = torch.empty(n, dtype=torch.uint8, device=dev)
    dst = torch.empty(n, dtype=torch.uint8, device=dev)
    for _ in range(3):
        dst.copy_(src)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(reps):
        e0.record()
        dst.copy_(src)
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


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
    traffic = traffic_for(f"tok_generate_observations_{rows}_rows")
    floor = stream_floor_ms(TOKENIZER_BYTES_PER_ROW * rows, dev)
    return {"kernel": "tok_generate_observations", "bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": ach / PEAK_HBM_GBS, "traffic": traffic, "avg_launch_ms": ms / n, "launches": n, "rows_per_launch": rows,
            "same_bytes_copy_ms": floor, "frac_of_same_bytes_copy": floor / (ms / n),
            "algorithmic_bytes_per_row": TOKENIZER_BYTES_PER_ROW,
            "note": "694 B/row = SURVEY 8d (u16 window 242 + own record 14 + 13 neighbour records 182 + uint8 row 256); the kernel "
                    "itself reads one-byte fields when every distance fits (573 B/row by its own layout)"}


def cfg4_instances(lo, hi, n_agents):
    """cfg4's instances lo..hi-1: every instance has its own synthetic 40 x 40 map (even ids Bernoulli "random" with obstacle
    density U[0.1, 0.3], odd ids "maze"), seeded by the GLOBAL instance id -> (grids u8 [n,50,50], pos, goal int16 [n,agents,2])."""
    from mapf_gpt_amd import maps
    gl, pl, gll = [], [], []
    for i in range(lo, hi):
        rng = np.random.Generator(np.random.PCG64([i, 4]))
        obst = maps.random_map(40, 40, float(rng.uniform(0.1, 0.3)), 1000 + i) if i % 2 == 0 else maps.maze_map(40, 40, 1000 + i)
        g = maps.pad(obst)
        p_, g_ = maps.place_agents(g, n_agents, i)
        gl.append(g); pl.append(p_); gll.append(g_)
    return np.stack(gl), torch.from_numpy(np.stack(pl)), torch.from_numpy(np.stack(gll))


def tokenizer_cfg4_launch(local_rank, reps=20):
    """The tokens kernel on cfg4's own per-GPU launch: 512 instances x 128 agents = 65 536 rows, every instance on its own
    map (the > 64-agent candidate-compaction path) -- the launch the north star's 8-GPU configuration actually issues."""
    from mapf_gpt_amd import _lib
    from mapf_gpt_amd.observation_generator import BatchedTokenizer
    dev = f"cuda:{local_rank}"
    _, n_agents, n_inst, _, _ = WORKLOADS["cfg4"]
    grid, pos, goal = cfg4_instances(0, n_inst, n_agents)
    pos, goal = pos.to(dev), goal.to(dev)
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
    floor = stream_floor_ms(TOKENIZER_BYTES_PER_ROW * rows, dev)
    return {"kernel": "tok_generate_observations", "bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": ach / PEAK_HBM_GBS, "traffic": traffic_for(f"cfg4_tok_generate_observations_{rows}_rows"),
            "avg_launch_ms": ms / n, "launches": n, "rows_per_launch": rows, "algorithmic_bytes_per_row": TOKENIZER_BYTES_PER_ROW,
            "same_bytes_copy_ms": floor, "frac_of_same_bytes_copy": floor / (ms / n),
            "note": "cfg4 per-GPU shard: 512 instances x 128 agents on per-instance 50 x 50 padded maps"}


def build_workload(name, precision, rank, world, local_rank, instances=0, use_graph=False, chunk_rows=0):
    """-> dict(run, pos, goal, grid, s_ok, g_ok, rows, n_total, ...) for this rank's shard of workload `name`."""
    from mapf_gpt_amd import maps
    from mapf_gpt_amd.model import build_model
    from mapf_gpt_amd.runner import BatchedRunner, make_instances, shard_range
    map_name, n_agents, inst_per_gpu, model, max_steps = WORKLOADS[name]
    if instances:
        inst_per_gpu = instances
    n_total = inst_per_gpu * world
    lo, hi = shard_range(n_total, rank, world)
    rows = (hi - lo) * n_agents
    # rows per forward launch.  85M: 4096 (round 6, tools/sweep_chunk85.sh: 1024 -> 1602, 2048 -> 1587, 4096 -> 1573 ms per cfg5 step; launches of
    # <= 512 rows are SLOWER -- the q|k|v and hidden planes of a small launch do not come back from the memory-side cache)
    chunk = min(rows, chunk_rows or (16384 if model != "85M" else 4096))
    # (MGPT_BENCH_ENVELOPE=ignore: timing-only ablation builds whose wrong logits the precision-envelope probe would answer with the fp32 kernels)
    net = build_model(model, seed=0, max_rows=chunk, precision=precision, device=f"cuda:{local_rank}",
                      envelope=os.environ.get("MGPT_BENCH_ENVELOPE", "fallback"))
    if name == "cfg4":                     # one map per instance, seeded by the global instance id
        grid, pos, goal = cfg4_instances(lo, hi, n_agents)
        s_ok = g_ok = None
    else:
        grid, s_ok, g_ok = maps.load_named(map_name)
        pos, goal = make_instances(grid, hi - lo, n_agents, first_seed=lo, start_ok=s_ok, goal_ok=g_ok)
    run = BatchedRunner(grid, hi - lo, n_agents, net, max_episode_steps=max_steps, seed=0, do_sample=True,
                        precision=precision, device=f"cuda:{local_rank}", row_offset=lo * n_agents, use_graph=use_graph)
    run.reset(pos, goal)
    return dict(name=name, run=run, net=net, pos=pos, goal=goal, grid=grid, s_ok=s_ok, g_ok=g_ok, rows=rows, n_total=n_total, chunk=chunk,
                n_agents=n_agents, inst_per_gpu=inst_per_gpu, model=model, max_steps=max_steps, map_name=map_name)


def timed_steps(w, steps, warmup, world, use_prof, coll_dev, sample_clock=None, collective=None):
    """W untimed warmup steps, then exactly `steps` steps between barrier + synchronize on both sides; max over ranks.
    collective: the process-group calls are made (default: world > 1; main() forces them at world = 1 under MGPT_BENCH_FORCE_COLLECTIVE)."""
    from mapf_gpt_amd import _lib
    run, pos, goal, max_steps = w["run"], w["pos"], w["goal"], w["max_steps"]
    collective = (world > 1) if collective is None else collective
    if collective:
        import torch.distributed as dist

    def barrier():
        torch.cuda.synchronize()
        if collective:
            dist.barrier()
        torch.cuda.synchronize()

    def hot_steps(k):
        for _ in range(k):
            if run.t >= max_steps:               # episode over: new episode (part of the job, stays inside the timing)
                run.reset(pos, goal)
            run.step()

    hot_steps(warmup)
    if use_prof:
        _lib.prof_reset()
        _lib.prof_enable(True)
    barrier()
    sampler = ClockPowerSampler(index=sample_clock) if sample_clock is not None else None
    if sampler:
        sampler.__enter__()
    t0 = time.perf_counter()
    hot_steps(steps)
    barrier()
    dt = time.perf_counter() - t0
    if sampler:
        sampler.__exit__()
        w["clock_power"] = sampler.summary(os.environ.get("MGPT_BENCH_CLOCK_LOG"))
    prof = {}
    if use_prof:
        _lib.prof_enable(False)
        prof = _lib.prof_read()
    w["local_dt"] = dt                           # this rank's own wall time between its two barriers (per_rank below)
    if collective:
        tt = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    return dt, prof


def per_rank_records(w, steps, rank, world, local_rank, coll_dev, collective):
    """One record per rank -- {rank, ms_per_step, sclk_mhz_mean, socket_power_w_mean, pci} -- so that an N-GPU line says on its face
    whether a shortfall was clocks, binding (two ranks on one PCI address) or a straggler.  One tiny all_gather
    of 8 float64 per rank, outside the timed region (the job's data path has no collective)."""
    cp = w.get("clock_power") or {}
    pr = torch.cuda.get_device_properties(local_rank)
    mine = [float(rank), 1e3 * w["local_dt"] / steps, float(cp.get("sclk_mhz_mean", float("nan"))), float(cp.get("socket_power_w_mean", float("nan"))),
            float(getattr(pr, "pci_domain_id", 0)), float(pr.pci_bus_id), float(pr.pci_device_id), float(local_rank)]
    rows = [mine]
    if collective:
        import torch.distributed as dist
        t = torch.tensor(mine, dtype=torch.float64, device=coll_dev)
        out = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(out, t)
        rows = [o.cpu().tolist() for o in out]
    recs = [{"rank": int(r[0]), "ms_per_step": r[1], "sclk_mhz_mean": (None if r[2] != r[2] else r[2]),
             "socket_power_w_mean": (None if r[3] != r[3] else r[3]), "pci": "%04x:%02x:%02x.0" % (int(r[4]), int(r[5]), int(r[6])),
             "local_rank": int(r[7])} for r in rows]
    return recs


def reset_cost(w, reps=3):
    """The episode boundary, which the timed region of a default run never crosses (steps < max_episode_steps): env reset + tokenizer
    create_agents (record init, one BFS distance field per agent: bfs_kernel, greedy bits) -- observation_generator.cpp:391-410 --
    HIP-event timed on the run's stream, outside the headline.  -> ms per episode of this rank's shard."""
    run, pos, goal = w["run"], w["pos"], w["goal"]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ms = []
    for _ in range(reps):
        torch.cuda.synchronize()
        e0.record()
        run.reset(pos, goal)
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    return float(min(ms))


def class_flops(prof, f_class, L_, rows, full=None):
    """Algorithmic (reference-executed) flops per step of every kernel class that actually ran; fused classes carry the
    flops of everything they absorbed.  The last-layer shortcut is OUR saving: flops stay the reference's.  `full` = the
    classes that ran FULL-SIZE launches (fold_last): a class that only ran for the last layer's token-255 rows (the 6M shape's
    out-projection since attn256o_kernel / attn256q_kernel absorbed it everywhere else) has been absorbed, its flops belong to the attention class."""
    ran = full if full is not None else prof
    cls = {k: f_class[k] * L_ * rows for k in f_class if k in prof and k in ran}
    if "gpt_attention" in cls:
        if "gpt_gemm_qkv" not in ran and "gpt_ln_qkv_fused" not in ran:
            cls["gpt_attention"] += f_class["gpt_gemm_qkv"] * L_ * rows
        if "gpt_gemm_attn_proj" not in ran:
            cls["gpt_attention"] += f_class["gpt_gemm_attn_proj"] * L_ * rows
    return cls


def roofline_of(prof, name, precision, model, rows, steps, rows_per_launch=None):
    """Dominant kernel class of a timed run (by HIP-event time) -> its roofline dict (algorithmic flops / event time)."""
    from mapf_gpt_amd import weights
    margs = weights.model_args(model)
    _, f_class = flops_per_row(margs)
    prof, full, last = fold_last(prof)
    cls = class_flops(prof, f_class, margs["n_layer"], rows, full)
    if not cls:
        return None
    dom = max(cls, key=lambda k: prof[k][0])
    ms, n = prof[dom]
    ach = cls[dom] * steps / (ms * 1e-3) / 1e12
    out = {"kernel": dom, "bound": "mfma", "achieved": ach, "peak": PEAK_TFLOPS[precision], "unit": "TFLOP/s",
           "frac": ach / PEAK_TFLOPS[precision], "avg_launch_ms": ms / n, "launches": n,
           "algorithmic_gflop_per_launch": cls[dom] * steps / n / 1e9, "traffic": traffic_for(f"{name}_{precision}_{dom}", rows_per_launch),
           "kernel_ms_per_step": {k: v[0] / steps for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])[:6]}}
    if rows_per_launch:
        out.update(full_launch_frac(dom, full, cls[dom] / (margs["n_layer"] * rows), rows_per_launch, PEAK_TFLOPS[precision]) or {})
        out["rows_per_launch"] = rows_per_launch
    return out


def secondary_shard(name, precision, steps, warmup, local_rank, coll_dev, instances=0):
    """A BASELINE configuration's per-GPU shard under the same clock as the headline: W warmup + exactly `steps` timed steps
    (barrier + synchronize on both sides), HIP-event hooks on -> value, ms/step and the dominant kernel's roofline."""
    w = build_workload(name, precision, 0, 1, local_rank, instances)
    dt, prof = timed_steps(w, steps, warmup, 1, True, coll_dev)
    out = {"workload": f"{name} per-GPU shard: {w['map_name']}, {w['n_agents']} agents, MAPF-GPT-{w['model']} shape, "
                       f"{w['inst_per_gpu']} instances, {w['rows']} rows/step",
           "value": w["n_total"] * w["n_agents"] * steps / dt, "unit": "agent-steps/s", "ms_per_step": 1e3 * dt / steps,
           "steps": steps, "warmup": warmup, "dtype": precision,
           "roofline": roofline_of(prof, name, precision, w["model"], w["rows"], steps, w["chunk"])}
    del w
    torch.cuda.empty_cache()
    return out


TRAFFIC_FILE = "r06_hbm_traffic.json"


def traffic_for(kernel_key, rows_per_launch=None):
    """HBM bytes per launch from THIS round's committed PMC passes (profiles/r06_hbm_traffic.json; no fallback to earlier
    rounds' files -- VERDICT r03: a stale entry is worse than null): a replay of a rocprofv3 run of this command, NOT a
    measurement of this run.  An entry that records the launch size it was measured at is only used for launches of that size."""
    tf = os.path.join(ROOT, "profiles", TRAFFIC_FILE)
    if not os.path.exists(tf):
        return None
    t = json.load(open(tf)).get(kernel_key)
    if not t or (rows_per_launch is not None and t.get("rows_per_launch", rows_per_launch) != rows_per_launch):
        return None
    return {"hbm_bytes_per_launch": t["fetch_corrected_x2"] + t["write"], "fetch_raw": t["fetch_raw"], "write": t["write"],
            "measured_in_run": False,
            "source": f"profiles/{TRAFFIC_FILE} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of this command, FETCH_SIZE x2 gfx950 correction)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="default: cfg3 at --gpus 1, cfg4 (its per-GPU shard) at --gpus N > 1")
    ap.add_argument("--precision", default=os.environ.get("MGPT_BENCH_PRECISION", "f16x3"), choices=["f32", "f16x3", "bf16"])
    ap.add_argument("--instances", type=int, default=0, help="instances per GPU (default: the workload's)")
    ap.add_argument("--chunk-rows", type=int, default=0, help="rows per forward launch (default 16384; 4096 for the 85M shape)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-tokenizer-leg", action="store_true", help="skip the >=1e5-row tokenizer roofline launch")
    ap.add_argument("--no-secondary", action="store_true", help="skip the short cfg2 run reported under 'secondary'")
    ap.add_argument("--no-prof", action="store_true", help="skip the per-kernel HIP-event hooks")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` typed without a launcher: become the launch line the contract names (one rank per GPU, rendezvous on
        # 127.0.0.1 -- the container hostname may not resolve); the ranks then run this file from the top with RANK / WORLD_SIZE set
        port = os.environ.get("MASTER_PORT", "29577")
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
                                  "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # MGPT_BENCH_BACKEND=gloo + MGPT_BENCH_SHARE_GPU=1: control-flow test of the N>1 path on a one-GPU box
    # (all ranks on cuda:0, collectives on CPU tensors); the real runs use RCCL ("nccl") with one GPU per rank.
    backend = os.environ.get("MGPT_BENCH_BACKEND", "nccl")
    if os.environ.get("MGPT_BENCH_SHARE_GPU"):
        local_rank = 0
    # MGPT_BENCH_FORCE_COLLECTIVE=1: take the process-group path (init, barriers, max-over-ranks all_reduce, the metrics all_gather)
    # at world = 1 too -- a one-rank RCCL communicator on a one-GPU box runs exactly the calls of the N > 1 job
    collective = world > 1 or bool(os.environ.get("MGPT_BENCH_FORCE_COLLECTIVE"))
    if collective:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_PORT", "29577")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        try:
            if backend == "nccl":
                local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
                assert torch.cuda.device_count() >= local_world or os.environ.get("MGPT_BENCH_SHARE_GPU"), \
                    f"{local_world} local ranks but only {torch.cuda.device_count()} visible GPUs (one rank per GPU)"
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)
        except Exception as e:                   # one diagnosable line per rank instead of a bare traceback from c10d
            print(f"[bench rank {rank}/{world}] init_process_group({backend}) failed: {type(e).__name__}: {e}; "
                  f"MASTER_ADDR={os.environ.get('MASTER_ADDR')} MASTER_PORT={os.environ.get('MASTER_PORT')} LOCAL_RANK={local_rank} "
                  f"visible_gpus={torch.cuda.device_count()} HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')}",
                  file=sys.stderr, flush=True)
            raise
        world = dist.get_world_size()            # the process group's own count is what n_gpus reports
    coll_dev = "cuda" if backend == "nccl" else "cpu"
    assert a.gpus == world, f"--gpus {a.gpus} but the process group has {world} ranks"
    torch.cuda.set_device(local_rank)

    from mapf_gpt_amd import weights
    from mapf_gpt_amd.runner import gather_metrics

    name = a.workload or ("cfg3" if world == 1 else "cfg4")
    w = build_workload(name, a.precision, rank, world, local_rank, a.instances, chunk_rows=a.chunk_rows)
    use_prof = not a.no_prof
    dt, prof_raw = timed_steps(w, a.steps, a.warmup, world, use_prof, coll_dev, sample_clock=local_rank, collective=collective)
    per_rank = per_rank_records(w, a.steps, rank, world, local_rank, coll_dev, collective)
    prof, prof_full, prof_last = fold_last(prof_raw)
    local_metrics = w["run"].metrics().to(coll_dev)
    metrics = gather_metrics(local_metrics, w["n_total"], rank, world, force=collective)       # the job's one collective
    gathered_on = str(metrics.device) if collective else None
    torch.cuda.synchronize()
    reset_ms = reset_cost(w) if rank == 0 else None          # (after the timed region and the metrics: the reset starts a new episode)
    eff_prec = None
    if rank == 0 and a.precision == "f16x3":
        try:
            eff_prec = w["net"].envelope()
        except Exception:
            eff_prec = None

    if rank == 0:
        model, n_agents, n_total, rows, map_name = w["model"], w["n_agents"], w["n_total"], w["rows"], w["map_name"]
        margs = weights.model_args(model)
        f_total, f_class = flops_per_row(margs)
        value = n_total * n_agents * a.steps / dt
        out = {"metric": "agent-steps/s (env+obs+GPT fwd)", "value": value, "unit": "agent-steps/s", "n_gpus": world,
               "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": a.precision, "data": "synthetic",
               "collective_backend": (backend if collective else None), "rccl_ranks": (world if collective and backend == "nccl" else None),
               "rccl_version": (".".join(map(str, torch.cuda.nccl.version())) if collective and backend == "nccl" else None),
               "metrics_gathered_on": gathered_on,
               "config": {"workload": f"{name}: {map_name}, {n_agents} agents, MAPF-GPT-{model} shape, "
                                      f"{w['inst_per_gpu']} instances/GPU ({n_total} total), {n_total * n_agents} rows/step",
                          "parallelism": f"instances sharded x{world}, no per-step collective",
                          "gflop_per_agent_step": f_total / 1e9,
                          "mean_ISR_after_run": float(metrics[:, 1].mean().item()),
                          "max_episode_steps": w["max_steps"],
                          "timed_region_crosses_episode_boundary": bool(a.warmup + a.steps > w["max_steps"]),
                          "reset_ms_per_episode": reset_ms,
                          "amortised_reset_ms_per_step": (reset_ms / w["max_steps"] if reset_ms is not None else None),
                          "reset_note": "env reset + tokenizer create_agents (one BFS field per agent, observation_generator.cpp:391-410) of this rank's "
                                        "shard, HIP-event timed once outside the headline; a default run's timed region ends before max_episode_steps"},
               "per_rank": per_rank,
               "effective_precision": ((eff_prec or {}).get("effective_precision") if a.precision == "f16x3" else a.precision),
               "precision_envelope": eff_prec,
               "note": "weights are seeded synthetic N(0, 0.02) tensors of the reference's shapes; the released checkpoints are "
                       "unreachable offline, so the 1e-5 logit parity of the f16x3 mode is established on synthetic weights "
                       "(tests/test_gpu_gpt.py) and not on the released ones; the f16x3 figures hold while the loaded checkpoint's precision "
                       "envelope stays 'inside' (effective_precision / precision_envelope: a checkpoint outside it is served by the fp32 kernels)"}
        if prof:
            cls = class_flops(prof, f_class, margs["n_layer"], rows, prof_full)
            if cls:
                dom = max(cls, key=lambda k: prof[k][0])
                ms, n = prof[dom]
                ach = cls[dom] * a.steps / (ms * 1e-3) / 1e12
                out["roofline"] = {"kernel": dom, "bound": "mfma", "achieved": ach, "peak": PEAK_TFLOPS[a.precision],
                                   "unit": "TFLOP/s", "frac": ach / PEAK_TFLOPS[a.precision],
                                   "traffic": traffic_for(f"{name}_{a.precision}_{dom}", w["chunk"]),
                                   "avg_launch_ms": ms / n, "launches": n, "rows_per_launch": w["chunk"],
                                   "algorithmic_gflop_per_launch": cls[dom] * a.steps / n / 1e9,
                                   "mfma_issue_frac": (3.0 if a.precision == "f16x3" else 1.0) * ach / PEAK_TFLOPS[a.precision],
                                   **(full_launch_frac(dom, prof_full, cls[dom] / (margs["n_layer"] * rows), w["chunk"], PEAK_TFLOPS[a.precision]) or {}),
                                   "note": "frac = algorithmic flops (reference-executed, all layers) / HIP-event time of the class over ALL its launches of the timed region, "
                                           "the last layer's token-255-only launches included; frac_full_launch = one layer's flops of a full launch / the mean duration of the full launches only"
                                           + ("; f16x3 issues 3 fp16 MFMAs per product against the fp16 dense peak: frac counts the reference's flops once, mfma_issue_frac counts the issued ones" if a.precision == "f16x3" else "")}
                out["roofline_all_classes"] = {k: {"tflops": cls[k] * a.steps / (prof[k][0] * 1e-3) / 1e12,
                                                   "frac": cls[k] * a.steps / (prof[k][0] * 1e-3) / 1e12 / PEAK_TFLOPS[a.precision],
                                                   "avg_launch_ms": prof[k][0] / prof[k][1]} for k in cls}
            if "tok_generate_observations" in prof:
                ms, n = prof["tok_generate_observations"]
                ach = TOKENIZER_BYTES_PER_ROW * rows / (ms / n * 1e-3) / 1e9
                out["roofline_tokenizer"] = {"kernel": "tok_generate_observations", "bound": "hbm", "achieved": ach,
                                             "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ach / PEAK_HBM_GBS,
                                             "traffic": traffic_for(f"{name}_tok_generate_observations"),
                                             "avg_launch_ms": ms / n, "launches": n, "rows_per_launch": rows,
                                             "note": "the workload's own launch (latency-bound when rows_per_launch < 1e5)"}
            out["kernel_ms_per_step"] = {k: v[0] / a.steps for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])}
            out["kernel_ms_per_step_last_layer_launches"] = {k: v[0] / a.steps for k, v in prof_last.items()}
        if w.get("clock_power"):
            out["clock_power"] = dict(w["clock_power"], note="sampled on this GPU between the two barriers of the timed region by a host thread reading hwmon every 20 ms -- the sampler runs DURING the timing (DESIGN section 10: the forward runs at the package power limit)")
        if world == 1 and not a.no_tokenizer_leg:
            # SURVEY 8d asks for the tokenizer's HBM roofline on >= 1e5-row launches: cfg4's own per-GPU launch (65 536 rows on
            # per-instance maps, 128 agents: the KP = 2 path) is the headline tokenizer figure; the 524 288-row leg is secondary
            out["roofline_tokenizer_cfg4_shard"] = tokenizer_cfg4_launch(local_rank)
            if name != "cfg4":
                out["roofline_tokenizer_large"] = tokenizer_large_launch(w["grid"], w["s_ok"], w["g_ok"], n_agents, local_rank)
        if world == 1 and not a.no_secondary and name != "cfg2":
            del w
            torch.cuda.empty_cache()
            w2 = build_workload("cfg2", a.precision, 0, 1, local_rank)
            dt2, _ = timed_steps(w2, 8, 2, 1, False, coll_dev)
            out["secondary"] = {"cfg2": {"workload": "cfg2: validation-mazes-seed-000, 64 agents, MAPF-GPT-2M shape, 256 instances, 16384 rows/step",
                                         "value": w2["n_total"] * w2["n_agents"] * 8 / dt2, "unit": "agent-steps/s", "ms_per_step": 1e3 * dt2 / 8,
                                         "steps": 8, "warmup": 2, "dtype": a.precision}}
            del w2
            torch.cuda.empty_cache()
            # cfg1 (the reference's own CPU-runnable case: one 32-agent instance): eager launches are the product path; the hipGraph replay of
            # the whole step is an option of BatchedRunner that does not pay here (the step is GPU-latency bound) and is timed beside it
            c1 = {}
            for tag, ug in (("eager", False), ("graph", True)):
                w1 = build_workload("cfg1", a.precision, 0, 1, local_rank, use_graph=ug)
                dt1, _ = timed_steps(w1, 120, 8, 1, False, coll_dev)
                c1[tag] = 1e3 * dt1 / 120
                del w1
            out["secondary"]["cfg1"] = {"workload": "cfg1: validation-random-seed-000, 32 agents, MAPF-GPT-2M shape, 1 instance, 32 rows/step",
                                        "value": 32 / (c1["eager"] * 1e-3), "unit": "agent-steps/s", "ms_per_step": c1["eager"],
                                        "ms_per_step_graph_replay": c1["graph"], "graph_speedup": c1["eager"] / c1["graph"],
                                        "steps": 120, "warmup": 8, "dtype": a.precision,
                                        "note": "value = eager launches (the default); the hipGraph replay of the step is an option (use_graph=True), timed beside it"}
            torch.cuda.empty_cache()
            # one 64-agent environment on the 6M model: launch-latency bound, eager launches
            w6 = build_workload("env6M", a.precision, 0, 1, local_rank)
            dt6, _ = timed_steps(w6, 60, 6, 1, False, coll_dev)
            out["secondary"]["env6M"] = {"workload": "one environment: validation-mazes-seed-000, 64 agents, MAPF-GPT-6M shape, 1 instance, 64 rows/step",
                                         "value": 64 * 60 / dt6, "unit": "agent-steps/s", "ms_per_step": 1e3 * dt6 / 60, "steps": 60, "warmup": 6,
                                         "dtype": a.precision,
                                         "note": "calls of <= 128 rows take the small-launch kernels of the 6M shape (head-parallel attention + packed out-projection, "
                                                 "64-token MLP blocks, one-launch last layer + head): DESIGN section 3.2"}
            del w6
            torch.cuda.empty_cache()
            # BASELINE configs[3] and [4]: the per-GPU shards of the two 8-GPU configurations, under this run's clock
            if name != "cfg4":
                out["secondary"]["cfg4_shard"] = secondary_shard("cfg4", "f16x3", 4, 1, local_rank, coll_dev)
                c4 = out["secondary"]["cfg4_shard"]
                # --gpus N > 1 runs cfg4 (its per-GPU shard): a 1 -> 8 curve must start from THIS number, not from cfg3's
                out["scale_reference"] = {"workload": c4["workload"], "value": c4["value"], "unit": "agent-steps/s", "ms_per_step": c4["ms_per_step"],
                                          "n_gpus": 1, "dtype": "f16x3",
                                          "note": "the N = 1 point of the multi-GPU (cfg4) curve; `bench.py --gpus 1 --workload cfg4` reproduces it as the headline"}
            if name != "cfg5":
                out["secondary"]["cfg5_shard"] = secondary_shard("cfg5", "bf16", 3, 1, local_rank, coll_dev)
        if world == 1 and not a.no_cpu_baseline and name != "cfg4":
            out["cpu_baseline"] = cpu_baseline(map_name, n_agents, model)
        print(json.dumps(out), flush=True)
    if collective:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
