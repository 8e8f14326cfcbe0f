#!/bin/bash
# tools/profile_r5b.sh -- round 5, after the packed GEMM changed: the default bench command under the kernel trace again, and the cfg5 counter passes
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD; OUT=gpurun_out/prof_r5; mkdir -p $OUT
(cd /tmp && timeout 1500 rocprofv3 --kernel-trace --stats -d $R/$OUT/kt -o bench -- python $R/bench.py > $R/$OUT/bench_under_rocprof.json 2> $R/$OUT/kt.log); echo "kt rc=$?"
python tools/rocpd_summary.py $OUT/kt/bench_results.db $OUT/bench_default_kernel_stats.txt | cut -c1-200 | head -24
bash tools/pmc_cfg5.sh
rm -rf $OUT/kt
