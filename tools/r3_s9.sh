#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r3s9; mkdir -p $OUT
timeout 180 tools/check_mlp256po $1 > $OUT/check_mlp256po.txt 2>&1; echo "check rc=$?"
cat $OUT/check_mlp256po.txt
