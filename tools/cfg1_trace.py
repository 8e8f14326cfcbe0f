#!/usr/bin/env python3
"""cfg1 (one 32-agent instance, 2M model) step anatomy.  Run under
    rocprofv3 --kernel-trace --output-format csv -d <dir> -- python tools/cfg1_trace.py run
then  python tools/cfg1_trace.py report <dir>  prints per-kernel mean duration, launches per step and the idle gaps between
consecutive kernels of a step (the part a graph / fewer launches can remove)."""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(use_graph=True, steps=200):
    import torch
    import bench
    w = bench.build_workload("cfg1", "f16x3", 0, 1, 0, use_graph=use_graph)
    w["run"].run(20)
    torch.cuda.synchronize()
    w["run"].run(steps)
    torch.cuda.synchronize()


def report(d, steps=200):
    f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[-(len(rows) * steps // (steps + 20)):]                     # drop warm-up
    dur, cnt, gap = {}, {}, 0
    busy = 0
    for a, b in zip(rows, rows[1:] + [None]):
        k = a["Kernel_Name"].split("(")[0][:70]
        dt = int(a["End_Timestamp"]) - int(a["Start_Timestamp"])
        dur[k] = dur.get(k, 0) + dt
        cnt[k] = cnt.get(k, 0) + 1
        busy += dt
        if b is not None:
            gap += max(0, int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
    span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
    print(f"{len(rows)} kernels, {len(rows) / steps:.1f} per step; span {span / steps / 1e3:.1f} us/step, busy {busy / steps / 1e3:.1f} us/step, "
          f"gaps {gap / steps / 1e3:.1f} us/step")
    for k in sorted(dur, key=lambda k: -dur[k]):
        print(f"{dur[k] / steps / 1e3:9.2f} us/step  {cnt[k] / steps:5.1f} launches/step  {dur[k] / cnt[k] / 1e3:8.2f} us each  {k}")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(use_graph=(len(sys.argv) < 3 or sys.argv[2] != "eager"))
    else:
        report(sys.argv[2])
