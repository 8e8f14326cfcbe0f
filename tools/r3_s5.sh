#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r3s5; mkdir -p $OUT
timeout 300 tools/probe_attn256 > $OUT/probe_attn256.txt 2>/dev/null; echo "probe rc=$?"
cat $OUT/probe_attn256.txt
