#!/usr/bin/env python3
"""tools/small_forward_latency.py -- forward latency of ONE environment's rows (32 agents = 32 rows of 256 tokens) through the three
released shapes: the way example.py uses the model (device-resident tokens in, logits out, launches only; no graph)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mapf_gpt_amd.model import build_model

for name, precision in (("2M", "f16x3"), ("6M", "f16x3"), ("85M", "f16x3"), ("2M", "bf16"), ("6M", "bf16"), ("85M", "bf16")):
    net = build_model(name, seed=0, max_rows=64, precision=precision)
    tok = torch.from_numpy(np.random.default_rng(0).integers(0, 67, (32, 256)).astype(np.uint8)).cuda()
    for _ in range(10):
        net.logits_tokens(tok)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100):
        net.logits_tokens(tok)
    torch.cuda.synchronize()
    print(f"{name:>4} {precision:>6}: 32 rows forward {1e3 * (time.perf_counter() - t0) / 100:8.3f} ms", flush=True)
    del net
