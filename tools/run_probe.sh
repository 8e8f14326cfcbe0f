#!/bin/bash
# tools/run_probe.sh <name> [args]  -- compile tools/bench_probes/<name>.hip for gfx950 if needed and run it on the GPU box
# (under `timeout`), output under gpurun_out/probes/<name>.txt.  e.g.  gpurun -- 'bash tools/run_probe.sh check_mlp256p'
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
N=$1; shift
OUT=gpurun_out/probes; mkdir -p $OUT
BIN=tools/bench_probes/$N
if [ ! -x $BIN ] || [ tools/bench_probes/$N.hip -nt $BIN ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -o $BIN tools/bench_probes/$N.hip || exit 1
fi
timeout 300 $BIN "$@" > $OUT/$N.txt 2>&1; echo "$N rc=$?"
cat $OUT/$N.txt
