#!/bin/bash
# cfg3 (6M, f16x3): rows per forward launch -- does a launch whose x (262 KB per row) fits the 256-MB memory-side cache run faster?
cd "$(dirname "$0")/.."
for n in ${CHUNKS:-12288 6144 3072 1536 768 12288}; do
  python bench.py --workload cfg3 --steps 10 --warmup 2 --chunk-rows $n --no-secondary --no-cpu-baseline --no-tokenizer-leg 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('chunk', $n, round(j['ms_per_step'],2), 'ms/step', round(j['value']), {k: round(v,2) for k,v in list(j['kernel_ms_per_step'].items())[:3]}, j['clock_power']['sclk_mhz_mean'], j['clock_power'].get('socket_power_w_mean'))"
done
