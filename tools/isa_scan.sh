#!/bin/bash
# usage: tools/isa_scan.sh [gwtf_bwd gwtf_stack ...]   (default: every csrc/*.hip)
# Compiles each source to gfx950 assembly and lists, per kernel, the memory instructions and exec-masked branches -- the check that found
# `cond ? *ptr : 0` compiled to one `s_and_saveexec / s_cbranch_execz / global_load_dword / s_or exec` block PER LOAD (docs/LOG.md, round 4):
# a kernel with about as many s_cbranch_execz as loads inside its hot loop has that pattern; scratch_* means spills or a local array
# that is indexed dynamically / captured by a lambda that was not inlined.
cd "$(dirname "$0")/../go_with_the_flows_amd/csrc"
SRCS=("$@"); [ ${#SRCS[@]} -eq 0 ] && SRCS=($(ls *.hip | sed 's/\.hip$//'))
for f in "${SRCS[@]}"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -fno-honor-nans \
    --cuda-device-only -S -o /tmp/isa_$f.s $f.hip 2>/dev/null || { echo "$f: compile failed"; continue; }
  python3 - /tmp/isa_$f.s $f <<'PY'
import re, sys, collections, subprocess
txt = open(sys.argv[1]).read()
for m in re.finditer(r'^(_Z\S+):\s*; @', txt, re.M):
    start = m.end()
    try: end = txt.index('s_endpgm', start)
    except ValueError: continue
    body = txt[start:end]
    c = collections.Counter(re.findall(r'^\s+(global_load_dword\b|global_load_dwordx[234]|global_load_lds_dwordx4|global_store_dword\b|global_store_dwordx[234]|'
                                       r'global_atomic_\w+|ds_add_f32|s_cbranch_execz|scratch_load_\w+|scratch_store_\w+|v_mfma_\w+)', body, re.M))
    name = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()
    name = re.sub(r'\(anonymous namespace\)::|^void ', '', name); name = re.sub(r'\(.*', '', name)
    total = len(re.findall(r'^\s+[a-z]\w+', body, re.M))
    print(f"{sys.argv[2]:22s} {name[:58]:58s} instr {total:5d}  " + ' '.join(f'{k}={v}' for k, v in sorted(c.items())))
PY
done
