#!/usr/bin/env python3
"""tools/condense_launch_series.py launch_series.txt out.txt -- the cfg3 launches (all 12 288 rows of a step per launch) of the three
6M kernels, first 240, from tools/launch_series.py's output; prints the medians."""
import statistics
import sys

lines = [l.rstrip() for l in open(sys.argv[1]) if l.strip()]
sel = []
for l in lines:
    p = l.split()
    g = int(p[2]); name = " ".join(p[3:])
    # round 6 names; rounds 3-5: mlp256p_kernel, attn256_kernel<F16T, 2, false (grid 6291456), gemm_pk_kernel<F16T, 2, 2 (grid 6291456)
    if ((name.startswith("mlp256q_kernel<F16T") or name.startswith("attn256q_kernel<F16T")) and g == 131072 and float(p[1]) > 2000) or \
       (name.startswith("attn_last1_kernel<256, 32, 4") and float(p[1]) > 1000):
        sel.append(" ".join(p[:3]) + " " + name)
sel = sel[:240]
open(sys.argv[2], "w").write("# per-launch durations (launch order) from the cfg3-only bench kernel trace: t_ms duration_us grid kernel\n"
                             "# cfg3 launches only (12 288 rows = one launch per layer and step): mlp256q_kernel / attn256q_kernel grid 131072 (256 persistent workgroups x 512), "
                             "attn_last1_kernel (last layer); first 240\n" + "\n".join(sel) + "\n")
for k in ("mlp256q", "attn256q", "attn_last1"):
    d = [float(s.split()[1]) for s in sel if k in s]
    if d:
        print(k, len(d), "median", round(statistics.median(d), 1), "min", round(min(d), 1), "max", round(max(d), 1))
