#!/bin/bash
# round-3 GPU visit 2: probe on realistic operands, with rocm-smi clock / power beside it
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r3s2; mkdir -p $OUT
(while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr -s ' ' | tr '\n' ';'; echo; sleep 0.15; done) > $OUT/smi_probe.txt &
SP=$!
timeout 300 tools/probe_mlp256 > $OUT/probe_mlp256.txt 2>&1; echo "probe rc=$?"
kill $SP
cat $OUT/probe_mlp256.txt | tail -8
grep -o "sclk[^;]*;[^;]*Power[^;]*" $OUT/smi_probe.txt | sed 's/ clock level: [0-9]*//' | awk '{print}' | sort | uniq -c | sort -rn | head -30
