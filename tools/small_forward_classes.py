#!/usr/bin/env python3
"""tools/small_forward_classes.py -- per-kernel-class time of a 32-row forward of the 6M shape (library HIP-event hooks): attention 7 x 117 us,
MLP 7 x 66 us, last layer 33 + 62 us, head 22 us (round 4) -- this shape has no small-launch kernels of its own."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mapf_gpt_amd.model import build_model
from mapf_gpt_amd import _lib
cases = [("6M", "f16x3")] if len(sys.argv) < 3 else [(sys.argv[1], sys.argv[2])]
for name, precision in cases:
    print(f"# {name} {precision}, 32 rows")
    net = build_model(name, seed=0, max_rows=64, precision=precision)
    tok = torch.from_numpy(np.random.default_rng(0).integers(0, 67, (32, 256)).astype(np.uint8)).cuda()
    for _ in range(5): net.logits_tokens(tok)
    _lib.prof_enable(True); _lib.prof_reset()
    for _ in range(50): net.logits_tokens(tok)
    r = _lib.prof_read(); _lib.prof_enable(False)
    for k,(ms,n) in sorted(r.items(), key=lambda kv:-kv[1][0]): print(f"{k:28s} {ms/50*1e3:9.1f} us/forward  {n/50:5.1f} launches  {ms/n*1e3:8.1f} us each")
