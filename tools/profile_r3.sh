#!/bin/bash
# tools/profile_r3.sh -- round-3 evidence run on the GPU box: kernel-trace stats of the default bench command, then separate
# counter passes (counters + kernel trace only, MI355X_MICROARCH.md HBM section): FETCH_SIZE, WRITE_SIZE, SQ busy, instruction mix.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD; OUT=gpurun_out/prof_r3; mkdir -p $OUT
(cd /tmp && timeout 1500 rocprofv3 --kernel-trace --stats -d $R/$OUT/kt -o bench -- python $R/bench.py > $R/$OUT/bench_under_rocprof.json 2> $R/$OUT/kt.log); echo "kt rc=$?"
python tools/rocpd_summary.py $OUT/kt/bench_results.db $OUT/kernel_stats.txt | cut -c1-200 | head -24
python tools/launch_series.py $OUT/kt/bench_results.db mlp256p_kernel attn256_kernel gemm_pk_kernel > $OUT/launch_series.txt
B="python $R/bench.py --no-cpu-baseline --no-secondary --steps 4 --warmup 1"
timeout 900 tools/pmc.sh FETCH_SIZE r3fetch $B | cut -c1-170
timeout 900 tools/pmc.sh WRITE_SIZE r3write $B | cut -c1-170
timeout 900 tools/pmc.sh "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" r3sq $B --no-tokenizer-leg --no-prof | cut -c1-250
timeout 900 tools/pmc.sh "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT" r3insts $B --no-tokenizer-leg --no-prof | cut -c1-250
