#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r3s7; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_tokenizer.py tests/test_gpu_parity_r2.py tests/test_gpu_loop.py tests/test_gpu_dataset_tokenizer.py -q -m gpu -p no:cacheprovider --timeout 600 -x -k "not 256_real and not bf16 and not heavy" > $OUT/pytest_tok.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_tok.log
timeout 600 python bench.py --no-cpu-baseline --no-secondary --steps 3 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3s7/bench.json").read().strip().splitlines()[-1])
for k in ("roofline_tokenizer", "roofline_tokenizer_cfg4_shard", "roofline_tokenizer_large"):
    v = d[k]; print(k, "rows", v["rows_per_launch"], "ms", round(v["avg_launch_ms"], 5), "frac", round(v["frac"], 3), "copy_ms", v.get("same_bytes_copy_ms"), "frac_of_copy", v.get("frac_of_same_bytes_copy"))
PY
python tools/bench_tokenizer.py 2>&1 | tail -12
