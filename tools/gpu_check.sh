#!/bin/bash
# tools/gpu_check.sh [quick]  -- everything one GPU-box visit should tell us, written under gpurun_out/
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out; mkdir -p $OUT
echo "== device ==" | tee $OUT/summary.txt
(rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -8; rocm-smi --showmeminfo vram | tail -4; nproc) >> $OUT/summary.txt 2>&1
echo "== smoke ==" | tee -a $OUT/summary.txt
timeout 600 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/summary.txt; tail -3 $OUT/smoke.log >> $OUT/summary.txt
echo "== pytest -m gpu ==" | tee -a $OUT/summary.txt
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 600 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt
tail -40 $OUT/pytest_gpu.log >> $OUT/summary.txt
echo "== bench ==" | tee -a $OUT/summary.txt
timeout 900 python bench.py --steps 8 --warmup 2 > $OUT/bench.log 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/summary.txt
cat $OUT/bench.log >> $OUT/summary.txt; tail -5 $OUT/bench.err >> $OUT/summary.txt
if [ "$1" != "quick" ]; then
  echo "== rocprofv3 kernel trace ==" | tee -a $OUT/summary.txt
  R=$PWD
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o bench -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-prof --no-tokenizer-leg > $R/$OUT/prof.log 2>&1); echo "rocprof rc=$?" | tee -a $OUT/summary.txt
  python tools/rocpd_summary.py $OUT/prof/bench_results.db $OUT/kernel_stats.txt | cut -c1-210 | head -24 >> $OUT/summary.txt
fi
cat $OUT/summary.txt
