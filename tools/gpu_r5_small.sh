#!/bin/bash
# tools/gpu_r5_small.sh -- round 5: small-launch path of the 6M shape: parity test, 32-row forward latency and its per-class split
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r5small; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_gpt.py -q -m gpu -x -p no:cacheprovider --timeout 600 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
timeout 600 python tools/small_forward_latency.py 2>/dev/null | tee $OUT/small_forward_latency.txt
timeout 600 python tools/small_forward_classes.py 2>/dev/null | tee $OUT/small_forward_classes.txt
