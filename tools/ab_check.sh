#!/bin/bash
# tools/ab_check.sh <libA> <libB> [workload precision steps]... -- the forward's parity tests under build B (gpurun_tmp/lib_B.so), then tools/ab_lib.sh A B
cd "$(dirname "$0")/.."
A=$1; B=$2; shift 2
cp mapf_gpt_amd/csrc/libmapf_gpt_amd.so /tmp/lib_keep.so
cp gpurun_tmp/lib_$B.so mapf_gpt_amd/csrc/libmapf_gpt_amd.so
python -m pytest tests/test_gpu_gpt.py tests/test_gpu_parity_r2.py tests/test_gpu_full_size.py tests/test_gpu_step_graph.py -m gpu -x -q 2>&1 | tail -3
cp /tmp/lib_keep.so mapf_gpt_amd/csrc/libmapf_gpt_amd.so
bash tools/ab_lib.sh $A $B "$@"
