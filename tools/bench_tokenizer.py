#!/usr/bin/env python3
"""tools/bench_tokenizer.py -- tokenizer kernels alone at large launches (GPU box): algorithmic GB/s per kernel."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from mapf_gpt_amd import _lib, maps
from mapf_gpt_amd.observation_generator import BatchedTokenizer
from mapf_gpt_amd.runner import make_instances

def run(map_name, n_inst, n_agents, reps=20):
    grid, s_ok, g_ok = maps.load_named(map_name)
    base = min(n_inst, 256)
    pos, goal = make_instances(grid, base, n_agents, 0, s_ok, g_ok)
    pos = pos.repeat((n_inst + base - 1) // base, 1, 1)[:n_inst].contiguous().cuda()
    goal = goal.repeat((n_inst + base - 1) // base, 1, 1)[:n_inst].contiguous().cuda()
    tok = BatchedTokenizer(grid, n_inst, n_agents)
    t0 = time.perf_counter(); tok.create_agents(pos, goal); torch.cuda.synchronize(); t_bfs = time.perf_counter() - t0
    act = torch.zeros((n_inst, n_agents), dtype=torch.int32, device="cuda")
    out = torch.empty((n_inst * n_agents, 256), dtype=torch.uint8, device="cuda")
    for _ in range(3):
        tok.update_agents(pos, goal, act, goals_may_change=False); tok.generate_observations(out)
    _lib.prof_reset(); _lib.prof_enable(True)
    for _ in range(reps):
        tok.update_agents(pos, goal, act, goals_may_change=False); tok.generate_observations(out)
    _lib.prof_enable(False)
    p = _lib.prof_read()
    rows = n_inst * n_agents
    ms = p["tok_generate_observations"][0] / reps
    ms_u = p["tok_update_agents"][0] / reps
    print(f"{map_name:28s} inst {n_inst:5d} x {n_agents:3d} = {rows:7d} rows | tokens {ms*1e3:8.1f} us  {694*rows/ms/1e6:8.1f} GB/s alg ({694*rows/ms/1e6/80:5.1f}% of 8 TB/s)"
          f" | update {ms_u*1e3:6.1f} us | create(BFS) {t_bfs*1e3:7.1f} ms", flush=True)

if __name__ == "__main__":
    run("validation-mazes-seed-000", 256, 64)
    run("validation-mazes-seed-000", 2048, 64)
    run("validation-mazes-seed-000", 8192, 64)
    run("validation-mazes-seed-000", 16384, 64)
    run("wfi_warehouse", 1024, 192)
    run("Berlin_1_256_00", 512, 256)
