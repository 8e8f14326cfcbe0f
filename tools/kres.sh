#!/bin/bash
# tools/kres.sh file.hip -- per-kernel register / scratch / LDS / occupancy table (hipcc remarks)
cd "$(dirname "$0")/../mapf_gpt_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$1" -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 | \
python3 -c "
import sys,re
cur=None; rows=[]
for line in sys.stdin:
    m=re.search(r'remark: (.*?) \[-Rpass', line)
    if not m: continue
    t=m.group(1).strip()
    if t.startswith('Function Name:'): cur={'name':t.split(': ',1)[1]}; rows.append(cur)
    elif cur is not None and ':' in t:
        k,v=t.split(':',1); cur[k.strip()]=v.strip()
import subprocess
for r in rows:
    name=subprocess.run(['c++filt',r['name']],capture_output=True,text=True).stdout.strip()[:90]
    print(f\"{name:90s} V={r.get('VGPRs')} A={r.get('AGPRs')} S={r.get('SGPRs')} scratch={r.get('ScratchSize [bytes/lane]')} occ={r.get('Occupancy [waves/SIMD]')} lds={r.get('LDS Size [bytes/block]')}\")
"
