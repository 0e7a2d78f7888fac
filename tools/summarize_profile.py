#!/usr/bin/env python3
"""Condense a tools/profile.sh output directory into the text summary committed under profiles/."""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
for w in ('airplane', 'm1'):
    print(f'===== workload {w} =====')
    st = glob.glob(os.path.join(root, f'trace_{w}', '*kernel_stats.csv'))
    if st:
        print('-- rocprofv3 --kernel-trace --stats (top kernels; __amd_rocclr_copyBuffer = start-up uploads of the synthetic parameters, '
              'one per tensor, none inside a step) --')
        for i, r in enumerate(csv.DictReader(open(st[0]))):
            if i < 6:
                print(f"{r['Name'][:90]:90s} calls={r['Calls']:>6s} avg_ns={float(r['AverageNs']):>12.1f} pct={r['Percentage']}")
    agg = collections.defaultdict(list)
    meta = {}
    for d in ('pmc_sq', 'pmc_mfma', 'pmc_fetch', 'pmc_write'):
        for f in glob.glob(os.path.join(root, f'{d}_{w}', '*counter_collection.csv')):
            for r in csv.DictReader(open(f)):
                if 'stack_kernel' in r['Kernel_Name']:
                    agg[r['Counter_Name']].append(float(r['Counter_Value']))
                    meta = {k: r[k] for k in ('Grid_Size', 'Workgroup_Size', 'LDS_Block_Size', 'VGPR_Count', 'Accum_VGPR_Count', 'SGPR_Count')}
    if agg:
        print('-- PMC, stack_kernel, mean per dispatch --', meta)
        for k in sorted(agg):
            print(f'{k:28s} {sum(agg[k]) / len(agg[k]):16.1f}   (n={len(agg[k])})')
        g = lambda k: sum(agg[k]) / len(agg[k]) if agg.get(k) else float('nan')
        waves = float(meta['Grid_Size']) / 64
        print(f'waves={waves:.0f}  VALU/wave={g("SQ_INSTS_VALU") / waves:.0f}  MFMA/wave={g("SQ_INSTS_MFMA") / waves:.0f}')
        if agg.get('SQ_VALU_MFMA_BUSY_CYCLES') and agg.get('GRBM_GUI_ACTIVE'):
            # SQ_VALU_MFMA_BUSY_CYCLES = matrix-pipe busy cycles summed over the 1024 SIMDs (= 16 x SQ_INSTS_MFMA for the
            # 16x16x32 f16 MFMA); GRBM_GUI_ACTIVE is summed over the 8 XCDs
            cyc = g('GRBM_GUI_ACTIVE') / 8
            print(f'kernel cycles ~ {cyc:.3e}; MFMA pipe utilisation = MFMA_BUSY / (cycles x 1024 SIMDs) = '
                  f'{g("SQ_VALU_MFMA_BUSY_CYCLES") / (cyc * 1024):.3f}; VALU issue share = 4 x SQ_INSTS_VALU / (cycles x 1024) = '
                  f'{4 * g("SQ_INSTS_VALU") / (cyc * 1024):.3f}')
        print(f'HBM bytes per dispatch: read = 2*FETCH_SIZE*1024 = {2 * g("FETCH_SIZE") * 1024:.3e} (gfx950 correction x2), '
              f'write = WRITE_SIZE*1024 = {g("WRITE_SIZE") * 1024:.3e}')


# ---- traffic.json: HBM-side bytes per stack_kernel launch for every profiled workload (bench.py's roofline.traffic) ----
import json
tag = sys.argv[2] if len(sys.argv) > 2 else 'untagged'
traffic = {'_tag': tag, '_comment': 'HBM-side bytes per stack_kernel launch = 2*FETCH_SIZE*1024 (gfx950: FETCH_SIZE counts half of a wide '
           'coalesced stream, MI355X_MICROARCH.md HBM section) + WRITE_SIZE*1024; rocprofv3 --pmc, separate passes, the bench\'s own eager '
           'launches (tools/profile.sh)'}
for w in ('airplane', 'm1', 'ae', 'svr', 'k16'):
    vals = {}
    for d, c in (('pmc_fetch', 'FETCH_SIZE'), ('pmc_write', 'WRITE_SIZE')):
        xs = [float(r['Counter_Value']) for f in glob.glob(os.path.join(root, f'{d}_{w}', '*counter_collection.csv'))
              for r in csv.DictReader(open(f)) if 'stack_kernel' in r['Kernel_Name'] and r['Counter_Name'] == c]
        if xs:
            vals[c] = sum(xs) / len(xs)
    if len(vals) == 2:
        traffic[w] = int(round(2 * vals['FETCH_SIZE'] * 1024 + vals['WRITE_SIZE'] * 1024, -3))
json.dump(traffic, open(os.path.join(root, 'traffic.json'), 'w'), indent=1)
print('traffic.json:', {k: v for k, v in traffic.items() if not k.startswith('_')})
