#!/bin/bash
# tools/ab_lib.sh <nameA> <nameB> [workload precision steps]...  -- A/B of two builds of the library (gpurun_tmp/lib_<name>.so)
# on bench.py inside ONE box, alternating A B A B (box-to-box variation is +-3 %, larger than most kernel changes)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
A=$1; B=$2; shift 2
[ $# -eq 0 ] && set -- cfg3 f16x3 20
OUT=gpurun_out/ab; mkdir -p $OUT
cp mapf_gpt_amd/csrc/libmapf_gpt_amd.so /tmp/lib_current.so
while [ $# -ge 3 ]; do
  W=$1; P=$2; S=$3; shift 3
  for v in $A $B $A $B; do
    cp gpurun_tmp/lib_$v.so mapf_gpt_amd/csrc/libmapf_gpt_amd.so
    timeout 600 python bench.py --workload $W --precision $P --steps $S --warmup 2 --no-cpu-baseline --no-secondary --no-tokenizer-leg > $OUT/$v.json 2> $OUT/$v.err
    python - <<PY
import json
d = json.loads(open("$OUT/$v.json").read().strip().splitlines()[-1])
print("%-10s $W $P" % "$v", round(d["ms_per_step"], 2), {k: round(x, 2) for k, x in d["kernel_ms_per_step"].items() if x > 0.5}, {k: round(x, 3) for k, x in d.get("kernel_ms_per_step_last_layer_launches", {}).items()})
PY
  done
done | tee $OUT/ab.txt
cp /tmp/lib_current.so mapf_gpt_amd/csrc/libmapf_gpt_amd.so
