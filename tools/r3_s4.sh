#!/bin/bash
# round-3 GPU visit: full GPU test suite + default bench (no cpu baseline) with the current library
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r3s4; mkdir -p $OUT
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 600 -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.log
timeout 600 python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 4 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3s4/bench.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "value", d["value"]); print(d["kernel_ms_per_step"]); print(d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
PY
