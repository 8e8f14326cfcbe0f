#!/bin/bash
# tools/gpu_bench_quick.sh [pytest selection]  -- selected GPU tests, then the cfg3 bench line without the secondary / cpu legs
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/quick; mkdir -p $OUT
timeout 1500 python -m pytest ${@:-tests} -q -m gpu -p no:cacheprovider --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest_gpu.log
timeout 600 python bench.py --no-cpu-baseline --no-secondary --no-tokenizer-leg --steps 20 --warmup 4 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/quick/bench.json").read().strip().splitlines()[-1])
print("ms_per_step", round(d["ms_per_step"], 2), "value", round(d["value"]), {k: round(v, 2) for k, v in list(d["kernel_ms_per_step"].items())[:4]}, "frac", round(d["roofline"]["frac"], 4))
PY
