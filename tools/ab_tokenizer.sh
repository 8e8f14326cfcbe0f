#!/bin/bash
# A/B of two builds of the library on the tokenizer bench inside ONE box (box-to-box variation is +-3 %)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r3s8; mkdir -p $OUT
cp mapf_gpt_amd/csrc/libmapf_gpt_amd.so /tmp/lib_current.so
for v in head tiled head tiled; do
  cp gpurun_tmp/lib_$v.so mapf_gpt_amd/csrc/libmapf_gpt_amd.so
  echo "== $v"; python tools/bench_tokenizer.py 2>/dev/null | tail -6
done | tee $OUT/ab.txt
cp /tmp/lib_current.so mapf_gpt_amd/csrc/libmapf_gpt_amd.so
