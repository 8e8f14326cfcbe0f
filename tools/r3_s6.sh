#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r3s6; mkdir -p $OUT
rm -f gpurun_out/parity_records.jsonl
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -25 $OUT/pytest_gpu.log
timeout 900 python bench.py --steps 8 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -3 $OUT/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3s6/bench.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "value", d["value"])
for k, v in d.get("secondary", {}).items():
    print(k, v["ms_per_step"], v["value"], (v.get("roofline") or {}).get("kernel"), (v.get("roofline") or {}).get("frac"))
print(d["cpu_baseline"]["value"], d["cpu_baseline"]["sample"][:200])
PY
