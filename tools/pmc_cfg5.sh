#!/bin/bash
# tools/pmc_cfg5.sh -- FETCH_SIZE / WRITE_SIZE passes over a short cfg5 (85M, bf16) run: the traffic entry of secondary.cfg5_shard.roofline
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD; OUT=gpurun_out/prof_r5; mkdir -p $OUT
B="python $R/bench.py --workload cfg5 --precision bf16 --instances 8 --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-tokenizer-leg --no-prof"
timeout 1200 tools/pmc.sh FETCH_SIZE r5fetch5 $B | cut -c1-170 | head -12
timeout 1200 tools/pmc.sh WRITE_SIZE r5write5 $B | cut -c1-170 | head -12
cp gpurun_out/pmc_r5fetch5/summary.txt $OUT/pmc_r5fetch_cfg5.txt; cp gpurun_out/pmc_r5write5/summary.txt $OUT/pmc_r5write_cfg5.txt
