#!/usr/bin/env python3
"""tokens kernel on cfg4's per-GPU launch (and the 524 160-row launch): HIP-event time per launch, bench.py's own legs."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mapf_gpt_amd import maps
r = bench.tokenizer_cfg4_launch(0, reps=50)
print("cfg4 65536 rows: %.2f us  frac %.3f  copy %.2f us" % (1e3 * r["avg_launch_ms"], r["frac"], 1e3 * r["same_bytes_copy_ms"]))
grid, s_ok, g_ok = maps.load_named("wfi_warehouse")
r = bench.tokenizer_large_launch(grid, s_ok, g_ok, 192, 0, reps=30)
print("warehouse %d rows: %.2f us  frac %.3f" % (r["rows_per_launch"], 1e3 * r["avg_launch_ms"], r["frac"]))
for n_inst in (128, 256, 1024):
    import torch
    from mapf_gpt_amd import _lib
    from mapf_gpt_amd.observation_generator import BatchedTokenizer
    g, pos, goal = bench.cfg4_instances(0, n_inst, 128)
    tok = BatchedTokenizer(g, n_inst, 128, device="cuda:0")
    pos, goal = pos.cuda(), goal.cuda()
    tok.create_agents(pos, goal)
    act = torch.zeros((n_inst, 128), dtype=torch.int32, device="cuda")
    out = torch.empty((n_inst * 128, 256), dtype=torch.uint8, device="cuda")
    for _ in range(3):
        tok.update_agents(pos, goal, act, goals_may_change=False); tok.generate_observations(out)
    _lib.prof_reset(); _lib.prof_enable(True)
    for _ in range(30):
        tok.update_agents(pos, goal, act, goals_may_change=False); tok.generate_observations(out)
    _lib.prof_enable(False)
    ms, n = _lib.prof_read()["tok_generate_observations"]
    print("cfg4-style %d instances (%d rows): %.2f us" % (n_inst, n_inst * 128, 1e3 * ms / n))
