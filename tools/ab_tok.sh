#!/bin/bash
# tools/ab_tok.sh <A> <B>  -- tokenizer tests with the current library, then the tokens kernel's timing legs (tools/tok_cfg4_time.py) for two library
# builds (gpurun_tmp/lib_<name>.so) alternated inside ONE box
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/abtok; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_tokenizer.py tests/test_gpu_parity_r2.py::test_large_launch_kernel_vs_oracle tests/test_gpu_parity_r2.py::test_cfg4_workload_tokens_and_6M_logits tests/test_gpu_loop.py tests/test_gpu_full_size.py tests/test_gpu_dataset_tokenizer.py -q -m gpu -x -p no:cacheprovider --timeout 600 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest.log
cp mapf_gpt_amd/csrc/libmapf_gpt_amd.so /tmp/lib_current.so
for v in $1 $2 $1 $2; do
  cp gpurun_tmp/lib_$v.so mapf_gpt_amd/csrc/libmapf_gpt_amd.so
  echo "== $v"; timeout 600 python tools/tok_cfg4_time.py 2>/dev/null | tail -5
done | tee $OUT/ab_$1_$2.txt
cp /tmp/lib_current.so mapf_gpt_amd/csrc/libmapf_gpt_amd.so
