#!/bin/bash
# tools/clock_under_load.sh -- sample rocm-smi (sclk, power, temperature) every 0.2 s while the cfg3 bench runs
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/clock.txt}
mkdir -p "$(dirname $OUT)"
(python bench.py --steps 250 --warmup 4 --no-cpu-baseline --no-secondary --no-tokenizer-leg > /tmp/bench_clock.json 2>/dev/null) &
BP=$!
sleep 7
echo "# rocm-smi samples while python bench.py (cfg3, f16x3) runs; idle sample last" > $OUT
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|Power|Temperature \(Sensor (edge|junction)" | tr -s ' ' | tr '\n' ';' >> $OUT; echo >> $OUT
  sleep 0.2
done
sleep 2
echo "# idle:" >> $OUT
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr -s ' ' | tr '\n' ';' >> $OUT; echo >> $OUT
tail -c 300 /tmp/bench_clock.json | head -c 200 >> /dev/null
sort $OUT | uniq -c | sort -rn | head -12
