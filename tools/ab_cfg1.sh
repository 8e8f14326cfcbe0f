cd /root/repo
for v in notail tail notail tail; do cp gpurun_tmp/lib_$v.so mapf_gpt_amd/csrc/libmapf_gpt_amd.so; python bench.py --workload cfg1 --steps 400 --warmup 20 --no-cpu-baseline --no-secondary --no-tokenizer-leg --no-prof 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['ms_per_step'],4), round(d['value']))"; done
cp gpurun_tmp/lib_tail.so mapf_gpt_amd/csrc/libmapf_gpt_amd.so
python tools/bench_list_api.py 2>&1 | grep -v amdgpu | head -2
