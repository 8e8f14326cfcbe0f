#!/bin/bash
# tools/profile_r2.sh [pmc-only] -- round-2 evidence run on the GPU box: kernel-trace stats of the default bench command, then
# separate FETCH_SIZE / WRITE_SIZE counter passes (counters + kernel trace only, MI355X_MICROARCH.md HBM section).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD; OUT=gpurun_out/prof_r2; mkdir -p $OUT
if [ "$1" != "pmc-only" ]; then
  (cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats -d $R/$OUT/kt -o bench -- python $R/bench.py > $R/$OUT/bench_under_rocprof.json 2> $R/$OUT/kt.log); echo "kt rc=$?"
  python tools/rocpd_summary.py $OUT/kt/bench_results.db $OUT/kernel_stats.txt | cut -c1-200 | head -30
fi
timeout 1200 tools/pmc.sh FETCH_SIZE r2fetch python $R/bench.py --no-cpu-baseline --no-secondary --steps 4 --warmup 1 | cut -c1-160
timeout 1200 tools/pmc.sh WRITE_SIZE r2write python $R/bench.py --no-cpu-baseline --no-secondary --steps 4 --warmup 1 | cut -c1-160
