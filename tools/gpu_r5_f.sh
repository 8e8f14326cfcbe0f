#!/bin/bash
# tools/gpu_r5_f.sh <A> <B> [workload precision steps]... -- A/B of two library builds only
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
A=$1; B=$2; shift 2
bash tools/ab_lib.sh $A $B "$@" 2>&1 | tail -12
mkdir -p gpurun_out/r5f; cp gpurun_out/ab/ab.txt gpurun_out/r5f/ab_${A}_${B}.txt
