#!/bin/bash
# round-3 GPU visit 1: baseline on this box + where mlp256_kernel's in-situ time goes
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r3s1; mkdir -p $OUT
ls /sys/class/drm/card*/device/hwmon/hwmon*/ > $OUT/sysfs.txt 2>&1
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 600 -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
python tools/clock_sampler.py $OUT/clock_bench.txt 0.01 & SP=$!
timeout 600 python bench.py --no-cpu-baseline --no-secondary --steps 30 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
kill -INT $SP
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3s1/bench.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], d["kernel_ms_per_step"])
PY
python tools/clock_sampler.py $OUT/clock_probe.txt 0.01 & SP=$!
timeout 300 tools/probe_mlp256 > $OUT/probe_mlp256.txt 2>&1; echo "probe rc=$?"
kill -INT $SP
cat $OUT/probe_mlp256.txt
awk 'NR>1 && $2>1000 {n++; s+=$2; p+=$3} END {if (n) print "bench: loaded samples", n, "mean sclk", s/n, "mean W", p/n}' $OUT/clock_bench.txt
awk 'NR>1 && $2>1000 {n++; s+=$2; p+=$3} END {if (n) print "probe: loaded samples", n, "mean sclk", s/n, "mean W", p/n}' $OUT/clock_probe.txt
