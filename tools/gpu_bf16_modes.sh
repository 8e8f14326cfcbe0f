#!/bin/bash
# tools/gpu_bf16_modes.sh -- the bf16 one-pass mode (outside the 1e-5 bar) on cfg3 / cfg2 / cfg4's shard, and f16x3 on cfg4's shard
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/modes; mkdir -p $OUT
for spec in "cfg3 bf16 20" "cfg2 bf16 20" "cfg4 bf16 4" "cfg4 f16x3 4"; do
  set -- $spec
  timeout 600 python bench.py --workload $1 --precision $2 --steps $3 --warmup 2 --no-cpu-baseline --no-secondary --no-tokenizer-leg > $OUT/$1_$2.json 2> $OUT/$1_$2.err
  python - <<PY
import json
d = json.loads(open("$OUT/$1_$2.json").read().strip().splitlines()[-1])
print("$1 $2", round(d["ms_per_step"], 2), "ms/step", round(d["value"]), "agent-steps/s", {k: round(x, 2) for k, x in d["kernel_ms_per_step"].items() if x > 0.5}, "frac", round(d["roofline"]["frac"], 3))
PY
done
