cd /root/repo; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/sec
for w in cfg2 cfg5; do
  P=f16x3; S=6; [ $w = cfg5 ] && P=bf16 && S=2
  timeout 600 python bench.py --workload $w --precision $P --steps $S --warmup 1 --no-cpu-baseline --no-secondary --no-tokenizer-leg > gpurun_out/sec/$w.json 2> gpurun_out/sec/$w.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/sec/$w.json").read().strip().splitlines()[-1])
print("$w", round(d["ms_per_step"],2), round(d["value"]), {k: round(v,2) for k,v in d["kernel_ms_per_step"].items()}, d["roofline"]["kernel"], round(d["roofline"]["frac"],3), d["config"]["workload"])
PY
done
