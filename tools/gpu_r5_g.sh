#!/bin/bash
# tools/gpu_r5_g.sh <A> <B> -- round 5: the policy's GPU parity tests with the library as built, then A/B of two library builds on the 85M shard
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r5g; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_gpt.py tests/test_gpu_parity_r2.py tests/test_gpu_full_size.py -q -m gpu -x -p no:cacheprovider --timeout 600 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|Error|error" $OUT/pytest.log | tail -8
bash tools/ab_lib.sh $1 $2 cfg5 bf16 6 2>&1 | tail -6
cp gpurun_out/ab/ab.txt $OUT/ab_$1_$2.txt
