#!/bin/bash
# tools/ab_tok_libs.sh <libA> <libB> ... -- tools/tok_cfg4_time.py under several builds of the library (gpurun_tmp/lib_<name>.so), one box, A B A B
cd "$(dirname "$0")/.."
cp mapf_gpt_amd/csrc/libmapf_gpt_amd.so /tmp/lib_current.so
for rep in 1 2; do for v in "$@"; do
  cp gpurun_tmp/lib_$v.so mapf_gpt_amd/csrc/libmapf_gpt_amd.so
  echo "== $v"; python tools/tok_cfg4_time.py 2>&1 | grep -v amdgpu.ids | head -2
done; done
cp /tmp/lib_current.so mapf_gpt_amd/csrc/libmapf_gpt_amd.so
