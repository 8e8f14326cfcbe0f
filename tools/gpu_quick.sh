#!/bin/bash
# tools/gpu_quick.sh [pytest args]  -- GPU test suite (or a selection) + the list-API bench, output under gpurun_out/quick/
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/quick; mkdir -p $OUT
timeout 1500 python -m pytest ${@:-tests} -q -m gpu -p no:cacheprovider --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest_gpu.log
timeout 300 python tools/bench_list_api.py > $OUT/list_api.txt 2>&1; cat $OUT/list_api.txt | tail -4
