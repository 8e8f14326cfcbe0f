#!/bin/bash
# round-3 GPU visit: mlp256p_kernel check (correctness + time next to mlp256_kernel)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r3s3; mkdir -p $OUT
timeout 120 tools/check_mlp256p $1 > $OUT/check_mlp256p.txt 2>&1; echo "check rc=$?"
cat $OUT/check_mlp256p.txt
