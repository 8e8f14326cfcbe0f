#!/usr/bin/env python3
"""tools/bench_tokenizer_one.py -- the bench.py `roofline_tokenizer_large` launch alone (cfg2 map, 8192 x 64 = 524 288 rows),
for PMC passes: every tokens_kernel dispatch of this process has the same size."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_tokenizer as b
b.run("validation-mazes-seed-000", 8192, 64, reps=10)
