#!/bin/bash
# timing ablations of the fused MLP kernel (GPU box): prints gpt_mlp_fused ms/step per variant
# usage: tools/abl_mlp.sh [cfg2|cfg3]   (1 no GELU, 2 no weight streaming, 3 no c_proj MFMAs, 4 no c_fc MFMAs; results are WRONG for != 0)
cd "$(dirname "$0")/.."
W=${1:-cfg2}
for a in 0 1 2 3 4; do
  MGPT_MLP_ABL=$a python bench.py --workload $W --steps 4 --warmup 1 --precision f16x3 --no-cpu-baseline --no-tokenizer-leg 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.readline()); print('ABL=$a', 'mlp_fused ms/step', round(d['kernel_ms_per_step']['gpt_mlp_fused'],2), 'total', round(d['ms_per_step'],2))"
done
