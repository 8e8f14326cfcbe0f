#!/bin/bash
# tools/gpu_r5_e.sh -- round 5: gemm_pk probe, then the policy's GPU parity tests
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r5e; mkdir -p $OUT gpurun_out/probes
timeout 300 tools/bench_probes/probe_gemm_pk > gpurun_out/probes/probe_gemm_pk.txt 2>&1; echo "probe rc=$?"; cut -c1-250 gpurun_out/probes/probe_gemm_pk.txt
timeout 1200 python -m pytest tests/test_gpu_gpt.py tests/test_gpu_parity_r2.py tests/test_gpu_full_size.py -q -m gpu -x -p no:cacheprovider --timeout 600 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|Error|error" $OUT/pytest.log | tail -8
