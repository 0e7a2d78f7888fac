#!/usr/bin/env python3
"""Per-source-line instruction census of one kernel: which lines of the .hip / .h sources the VALU, MFMA, LDS and SALU instructions
of a compiled kernel come from (straight-line kernels: static count = executed count per wave up to the branches taken).

    hipcc <flags of csrc/Makefile> --cuda-device-only -gline-tables-only -S -o /tmp/k.s csrc/gwtf_bwd.hip
    python tools/isa_lines.py /tmp/k.s 'bwd_kernelILi3ELi2ELi3ELi1ELi1ELb1' [--top 40] [--ranges 196-233:recompute,...]
"""
import argparse, collections, re
ap = argparse.ArgumentParser()
ap.add_argument('asm'); ap.add_argument('kernel', help='substring of the mangled kernel name')
ap.add_argument('--top', type=int, default=40)
ap.add_argument('--ranges', default='', help='comma list lo-hi:label of lines of the MAIN source file to aggregate')
a = ap.parse_args()
files, cur, on = {}, None, False
per = collections.defaultdict(lambda: collections.Counter())
for ln in open(a.asm):
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', ln)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split('/')[-1]
        continue
    if re.match(r'^_Z\S+:', ln):
        on = a.kernel in ln
        continue
    if not on:
        continue
    if 's_endpgm' in ln:
        on = False
        continue
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', ln)
    if m:
        cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
        continue
    m = re.match(r'\s+([a-z]\w+)', ln)
    if not m or cur is None:
        continue
    op = m.group(1)
    kind = ('MFMA' if op.startswith('v_mfma') else 'VALU' if op.startswith('v_') else 'LDS' if op.startswith('ds_') else
            'SALU' if op.startswith('s_') else 'VMEM' if op.startswith(('global_', 'buffer_', 'scratch_', 'flat_')) else 'other')
    per[cur][kind] += 1
tot = collections.Counter()
for c in per.values():
    tot.update(c)
print('total', dict(tot))
rows = sorted(per.items(), key=lambda kv: -kv[1]['VALU'])
for (f, l), c in rows[:a.top]:
    print(f'{f}:{l:<5d} VALU={c["VALU"]:5d} MFMA={c["MFMA"]:4d} LDS={c["LDS"]:4d} SALU={c["SALU"]:4d} VMEM={c["VMEM"]:3d}')
if a.ranges:
    main = collections.Counter(f for (f, l) in per).most_common(1)[0][0]
    print('ranges of', main)
    for r in a.ranges.split(','):
        span, label = r.split(':')
        lo, hi = map(int, span.split('-'))
        c = collections.Counter()
        for (f, l), cc in per.items():
            if f == main and lo <= l <= hi:
                c.update(cc)
        print(f'  {label:24s} {lo}-{hi}: VALU={c["VALU"]} MFMA={c["MFMA"]} LDS={c["LDS"]} SALU={c["SALU"]} VMEM={c["VMEM"]}')
