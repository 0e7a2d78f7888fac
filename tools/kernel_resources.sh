#!/bin/bash
# usage: tools/kernel_resources.sh gwtf_prior [gwtf_heads ...] -- VGPRs / AGPRs / scratch / LDS / occupancy of every kernel of a csrc file (hipcc remarks)
cd "$(dirname "$0")/../go_with_the_flows_amd/csrc"
for f in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -fno-honor-nans \
    -Rpass-analysis=kernel-resource-usage -c $f.hip -o /tmp/kr_$f.o 2>&1 |
  python3 -c "
import sys,re,subprocess
rows=[];cur={}
for l in sys.stdin:
    m=re.search(r'remark:\s+(.*?)\s*\[-Rpass',l)
    if not m: continue
    t=m.group(1)
    if t.startswith('Function Name:'):
        if cur: rows.append(cur)
        cur={'name':t.split(':',1)[1].strip()}
    elif ':' in t:
        k,v=t.split(':',1); cur[k.strip()]=v.strip()
if cur: rows.append(cur)
for r in rows:
    n=subprocess.run(['c++filt',r['name']],capture_output=True,text=True).stdout.strip()
    n=re.sub(r'\(anonymous namespace\)::','',n); n=re.sub(r'\(.*','',n)
    print(f\"{n[:60]:60s} vgpr {r.get('VGPRs','?'):>4s} agpr {r.get('AGPRs','?'):>4s} scratch {r.get('ScratchSize [bytes/lane]','?'):>5s} lds {r.get('LDS Size [bytes/block]','?'):>6s} occ {r.get('Occupancy [waves/SIMD]','?')}\")
"
done
