#!/bin/bash
# cfg5 shard (85M, bf16): rows per forward launch -- do q|k|v / hidden planes of a smaller launch stay in the 256-MB memory-side cache?
cd "$(dirname "$0")/.."
for n in ${CHUNKS:-1024 512 256 192 128 1024}; do
  python bench.py --workload cfg5 --precision bf16 --steps 2 --warmup 1 --chunk-rows $n --no-secondary --no-cpu-baseline --no-tokenizer-leg 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('chunk', $n, round(j['ms_per_step'],1), 'ms/step', round(j['value']), {k: round(v,1) for k,v in list(j['kernel_ms_per_step'].items())[:7]})"
done
