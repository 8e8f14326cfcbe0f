#!/bin/bash
# tools/build_ab.sh <name> [extra hipcc flags]  -- build a variant of the library into gpurun_tmp/lib_<name>.so (for tools/ab_lib.sh);
# e.g.  tools/build_ab.sh unfused -DMGPT_AB_ATTN_UNFUSED ; tools/build_ab.sh new
cd "$(dirname "$0")/.."
N=$1; shift
B=/tmp/ab_build_$N; mkdir -p $B gpurun_tmp
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable -Wno-unused-but-set-variable $*"
pids=()
for s in prof tokenizer env gpt gpt_fast step; do
  /opt/rocm/bin/hipcc $FLAGS -c mapf_gpt_amd/csrc/$s.hip -o $B/$s.o & pids+=($!)
done
for p in "${pids[@]}"; do wait $p || exit 1; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_tmp/lib_$N.so $B/*.o && ls -la gpurun_tmp/lib_$N.so
