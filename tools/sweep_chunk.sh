#!/bin/bash
# tools/sweep_chunk.sh -- cfg3 step time against the number of rows per forward launch (x of a launch = rows MiB: does the
# residual stream of one launch fit the 256 MB Infinity Cache?)
cd "$(dirname "$0")/.."
for c in "$@"; do
  timeout 300 python bench.py --chunk-rows $c --no-cpu-baseline --no-secondary --no-tokenizer-leg 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('chunk', $c, 'agent-steps/s', round(d['value']), 'ms/step', round(d['ms_per_step'], 1), {k: round(v, 1) for k, v in list(d['kernel_ms_per_step'].items())[:4]})"
done
