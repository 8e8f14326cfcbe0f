#!/bin/bash
# tools/profile_r6.sh -- round-6 evidence run on the GPU box (as tools/profile_r5.sh; the tokenizer's instruction counts are tools/tok_pmc.sh).
#   1. kernel-trace stats of the cfg3-ONLY bench command (one workload per stats file: avg x launches recomputes roofline.frac)
#   2. kernel-trace stats of the default bench command (secondary workloads and tokenizer legs included)
#   3. separate counter passes (counters + kernel trace only): FETCH_SIZE, WRITE_SIZE, SQ busy, instruction mix; cfg5 FETCH / WRITE
#   4. the bench's own clock / power log
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD; OUT=gpurun_out/prof_r6; mkdir -p $OUT
C3="python $R/bench.py --no-secondary --no-cpu-baseline --steps 10 --warmup 2"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/$OUT/kt3 -o cfg3 -- $C3 > $R/$OUT/cfg3_under_rocprof.json 2> $R/$OUT/kt3.log); echo "kt3 rc=$?"
python tools/rocpd_summary.py $OUT/kt3/cfg3_results.db $OUT/cfg3_kernel_stats.txt | cut -c1-200 | head -14
python tools/launch_series.py $OUT/kt3/cfg3_results.db mlp256q_kernel attn256q_kernel attn_last1_kernel > $OUT/launch_series.txt
(cd /tmp && timeout 1500 rocprofv3 --kernel-trace --stats -d $R/$OUT/kt -o bench -- python $R/bench.py > $R/$OUT/bench_under_rocprof.json 2> $R/$OUT/kt.log); echo "kt rc=$?"
python tools/rocpd_summary.py $OUT/kt/bench_results.db $OUT/bench_default_kernel_stats.txt | cut -c1-200 | head -24
B="python $R/bench.py --no-cpu-baseline --no-secondary --steps 4 --warmup 1"
timeout 900 tools/pmc.sh FETCH_SIZE r6fetch $B | cut -c1-170
timeout 900 tools/pmc.sh WRITE_SIZE r6write $B | cut -c1-170
timeout 900 tools/pmc.sh "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" r6sq $B --no-tokenizer-leg --no-prof | cut -c1-250
timeout 900 tools/pmc.sh "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT" r6insts $B --no-prof | cut -c1-250
B5="python $R/bench.py --workload cfg5 --precision bf16 --instances 16 --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-tokenizer-leg --no-prof"
timeout 1200 tools/pmc.sh FETCH_SIZE r6fetch5 $B5 | cut -c1-170 | head -12
timeout 1200 tools/pmc.sh WRITE_SIZE r6write5 $B5 | cut -c1-170 | head -12
MGPT_BENCH_CLOCK_LOG=$R/$OUT/clock_power_under_load.txt python bench.py --no-secondary --no-cpu-baseline --no-tokenizer-leg --steps 40 --warmup 4 > $OUT/bench_clock.json 2>/dev/null
head -3 $OUT/clock_power_under_load.txt; python -c "
import json; d=json.load(open('$OUT/bench_clock.json')); print(d.get('clock_power'))"
for t in r6fetch r6write r6sq r6insts; do cp gpurun_out/pmc_$t/summary.txt $OUT/pmc_$t.txt; done
cp gpurun_out/pmc_r6fetch5/summary.txt $OUT/pmc_r6fetch_cfg5.txt; cp gpurun_out/pmc_r6write5/summary.txt $OUT/pmc_r6write_cfg5.txt
python tools/make_hbm_traffic.py $OUT/pmc_r6fetch.txt $OUT/pmc_r6write.txt $OUT/hbm_traffic.json $OUT/pmc_r6fetch_cfg5.txt $OUT/pmc_r6write_cfg5.txt
rm -rf $OUT/kt $OUT/kt3
