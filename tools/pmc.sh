#!/bin/bash
# tools/pmc.sh "<counters>" <tag> [bench args...]  -- one rocprofv3 PMC pass over a short bench run (GPU box)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
CTRS="$1"; TAG="$2"; shift 2
R=$PWD
mkdir -p gpurun_out/pmc_$TAG
(cd /tmp && rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$TAG -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof "$@" > $R/gpurun_out/pmc_$TAG/log.txt 2>&1)
f=$(find gpurun_out/pmc_$TAG -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"][:70]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
disp = collections.Counter()
for r in rows:
    disp[(r["Kernel_Name"][:70], r["Dispatch_Id"])] += 0
for (k, d) in disp: cnt[k] += 1
names = sorted({r["Counter_Name"] for r in rows})
print("kernel".ljust(70), "disp", *[n[-22:].rjust(22) for n in names])
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", kv[1].get(names[0], 0)))[:12]:
    print(k.ljust(70), str(cnt[k]).rjust(4), *[("%.4g" % (v[n] / max(cnt[k], 1))).rjust(22) for n in names])
PY
