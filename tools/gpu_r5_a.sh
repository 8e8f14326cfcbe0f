#!/bin/bash
# tools/gpu_r5_a.sh -- round 5, visit A: parity tests of the policy kernels, A/B clumped vs interleaved attention loop, attention probe
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r5a; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_gpt.py tests/test_gpu_parity_r2.py tests/test_gpu_full_size.py -q -m gpu -x -p no:cacheprovider --timeout 600 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
bash tools/ab_lib.sh clumped new cfg3 f16x3 20 2>&1 | tail -8
cp gpurun_out/ab/ab.txt $OUT/ab_attn_loop.txt
bash tools/run_probe.sh check_attn256o 2>&1 | tail -8
