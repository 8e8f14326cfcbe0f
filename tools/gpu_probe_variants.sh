#!/bin/bash
# tools/gpu_probe_variants.sh <probe> <variant>...  -- run prebuilt variants tools/bench_probes/<probe>_<variant> one after the other in ONE box
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
P=$1; shift
OUT=gpurun_out/probes; mkdir -p $OUT
for v in "$@"; do
  B=tools/bench_probes/${P}_$v; [ "$v" = base ] && B=tools/bench_probes/$P
  echo "== $v"; timeout 200 $B > $OUT/${P}_$v.txt 2>&1; echo "rc=$?"; grep -v "^step phases" $OUT/${P}_$v.txt | cut -c1-400
done
