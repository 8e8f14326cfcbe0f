#!/bin/bash
# tools/run_8gpu.sh [N=8] [STEPS=20] [WARMUP=3] -- the exact multi-GPU command the driver issues (one rank per GPU over RCCL),
# N = 1, 2, 4, 8 back to back when called with "curve".  cfg4 (BASELINE configs[3]): 512 instances x 128 agents per GPU,
# instance ids sharded over ranks, no per-step collective, one all_gather of episode metrics after the run.
# The JSON line's `per_rank` (ms_per_step, shader clock, socket power and PCI address of every rank) says whether a curve below
# 8 x was clocks, rank <-> GPU binding or a straggler.
set -e
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0          # dmabuf IPC: RCCL across processes needs it on this driver
run() {
    local n=$1 steps=${2:-20} warm=${3:-3}
    if [ "$n" = 1 ]; then
        python bench.py --gpus 1 --workload cfg4 --steps "$steps" --warmup "$warm" --no-secondary --no-cpu-baseline
    else
        python -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 --master-port $((29500 + n)) \
            bench.py --gpus "$n" --steps "$steps" --warmup "$warm"
    fi
}
if [ "$1" = curve ]; then
    for n in 1 2 4 8; do run $n "${2:-20}" "${3:-3}"; done
else
    run "${1:-8}" "${2:-20}" "${3:-3}"
fi
