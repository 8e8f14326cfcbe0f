#!/bin/bash
# tools/tok_lds_ablation.sh -- what the tokens kernel's LDS pipe spends its cycles on: builds gpurun_tmp/lib_{base,abl1..5}.so (tools/build_ab.sh abl<N> -DMGPT_ABL_TOK=<N>;
# results are WRONG for N != 0), each timed and counted (SQ_LDS_IDX_ACTIVE, SQ_LDS_BANK_CONFLICT) on bench.py's two tokenizer launches.  One box.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
cp mapf_gpt_amd/csrc/libmapf_gpt_amd.so /tmp/lib_current.so
for v in base abl1 abl2 abl3 abl4 abl5 base; do
  cp gpurun_tmp/lib_$v.so mapf_gpt_amd/csrc/libmapf_gpt_amd.so
  echo "== $v"; python tools/tok_cfg4_time.py 2>&1 | grep -v amdgpu.ids | head -2
  bash tools/pmc.sh "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU" toklds_$v python $PWD/tools/tok_cfg4_time.py | grep -E "<3, 8>|=1048576"
done
cp /tmp/lib_current.so mapf_gpt_amd/csrc/libmapf_gpt_amd.so
