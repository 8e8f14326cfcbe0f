#!/bin/bash
# tools/gpu_r5_b.sh -- round 5, visit B: policy parity tests, A/B of the attention key-tile loops (clumped = round 4, runmax = pipelined with the
# running maximum, new = pipelined with one reference per head), attention probe
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r5b; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_gpt.py tests/test_gpu_parity_r2.py tests/test_gpu_full_size.py -q -m gpu -x -s -p no:cacheprovider --timeout 600 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "spread scores|passed|failed|Error|error" $OUT/pytest.log | tail -8
bash tools/ab_lib.sh ${1:-runmax} new cfg3 f16x3 20 2>&1 | tail -8
cp gpurun_out/ab/ab.txt $OUT/ab_attn_loop.txt
bash tools/gpu_probe_variants.sh check_attn256o ${2:-rm} base | grep -E "^==|per 12288|stamps" | cut -c1-330
