#!/bin/bash
# tools/pmc.sh "<counters>" <tag> <command...>  -- one rocprofv3 PMC pass (counters only + kernel trace) over a command,
# per-kernel per-dispatch averages printed and saved to gpurun_out/pmc_<tag>/summary.txt  (GPU box)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
CTRS="$1"; TAG="$2"; shift 2
R=$PWD
mkdir -p gpurun_out/pmc_$TAG
(cd /tmp && rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$TAG -o pmc -- "$@" > $R/gpurun_out/pmc_$TAG/log.txt 2>&1)
f=$(find gpurun_out/pmc_$TAG -name "*counter_collection.csv" | head -1)
python - "$f" "$CTRS" "$*" <<'PY' | tee gpurun_out/pmc_$TAG/summary.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for r in rows:
    k = (r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-60:] + " grid=" + r.get("Grid_Size", "?"))[:78]      # launches of different sizes stay apart
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
names = sorted({r["Counter_Name"] for r in rows})
print("# rocprofv3 --pmc %s --kernel-trace -- %s" % (sys.argv[2], sys.argv[3]))
print("# per-dispatch averages (MI355X)")
print("kernel".ljust(78), "disp", *[n[-22:].rjust(22) for n in names])
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1].values()))[:14]:
    n = len(disp[k]); print(k.ljust(78), str(n).rjust(4), *[("%.5g" % (v[c] / n)).rjust(22) for c in names])
PY
