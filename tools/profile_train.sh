# usage (GPU box): bash tools/profile_train.sh <tag>  -- kernel-trace stats + PMC passes of the train step (one 33-coupling
# airplane component, 64 x 2048 points: tools/diag/trainstep_kernels.py), per kernel.  Summary -> gpurun_out/prof_<tag>_train/summary.txt
TAG=${1:-r09}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_${TAG}_train; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ts -o ts -- python3 tools/diag/trainstep_kernels.py > $OUT/ts.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc1 -o ts -- python3 tools/diag/trainstep_kernels.py > $OUT/pmc1.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/pmc2 -o ts -- python3 tools/diag/trainstep_kernels.py > $OUT/pmc2.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -o ts -- python3 tools/diag/trainstep_kernels.py > $OUT/pmc3.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc4 -o ts -- python3 tools/diag/trainstep_kernels.py > $OUT/pmc4.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/full -o full -- python3 tools/bench_train.py --steps 5 > $OUT/full.log 2>&1
python3 tools/bench_train.py --graph --steps 10 > $OUT/full_graph.log 2>&1
python3 - > $OUT/summary.txt 2>&1 <<PY
import csv, glob, collections
out = "$OUT"
f = glob.glob(f"{out}/ts/*kernel_stats.csv")
rows = list(csv.DictReader(open(f[0]))) if f else []
rows = [r for r in rows if 'copyBuffer' not in r['Name']]
print("===== train step, one 33-coupling component (f=37), 64 x 2048 points, eager, 7 steps (tools/diag/trainstep_kernels.py) =====")
print(f"kernel time per step (start-up uploads excluded): {sum(float(r['TotalDurationNs']) for r in rows)/7/1e6:.2f} ms, kernels per step: {sum(int(r['Calls']) for r in rows)/7:.0f}")
for r in rows[:16]:
    print(f"  {r['Name'][:90]:90s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:9.2f} pct={r['Percentage']}")
print("===== whole airplane train step: K=4 x 33 couplings f=37, encoder, prior flow, mixture NLL, backward, fused AMSGrad (tools/bench_train.py) =====")
print("  " + "\n  ".join(l for l in open(f"{out}/full_graph.log").read().splitlines() if "ms/step" in l))
ff = glob.glob(f"{out}/full/*kernel_stats.csv")
if ff:
    fr = [r for r in csv.DictReader(open(ff[0])) if 'copyBuffer' not in r['Name'] and 'FillFunctor' not in r['Name']]
    print(f"  kernel time per eager step (start-up uploads and first-step optimiser state fills excluded): {sum(float(r['TotalDurationNs']) for r in fr)/7/1e6:.2f} ms, kernels per step: {sum(int(r['Calls']) for r in fr)/7:.0f}")
    for r in fr[:24]:
        print(f"  {r['Name'][:90]:90s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:9.2f} pct={r['Percentage']}")
agg = collections.defaultdict(lambda: collections.defaultdict(list)); meta = {}
for d in ("pmc1", "pmc2", "pmc3", "pmc4"):
    for fn in glob.glob(f"{out}/{d}/*counter_collection.csv"):
        for r in csv.DictReader(open(fn)):
            k = r["Kernel_Name"]
            key = next((n for n in ("bwd_kernel<3, 2, 3", "bwd_kernel<3, 2, 2", "bwd_kernel<3, 4, 2", "stats_kernel", "stack_kernel", "bwd_tail1_kernel", "fold1_bwd_kernel", "fold0_kernel", "fold1_kernel") if n in k), None)
            if key is None: continue
            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[key] = {m: r[m] for m in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size")}
for key in agg:
    a = agg[key]; g = lambda c: sum(a[c])/len(a[c]) if a.get(c) else float('nan')
    print(f"-- PMC {key}: mean per dispatch -- {meta[key]}")
    for c in sorted(a): print(f"   {c:28s} {g(c):16.1f} (n={len(a[c])})")
    waves = float(meta[key]["Grid_Size"]) / 64
    cyc = g("GRBM_GUI_ACTIVE") / 8
    print(f"   waves={waves:.0f} VALU/wave={g('SQ_INSTS_VALU')/waves:.0f} MFMA/wave={g('SQ_INSTS_MFMA')/waves:.0f} LDS/wave={g('SQ_INSTS_LDS')/waves:.0f} "
          f"kernel cycles~{cyc:.3e}  MFMA pipe busy={g('SQ_VALU_MFMA_BUSY_CYCLES')/(cyc*1024):.3f}  VALU issue={4*g('SQ_INSTS_VALU')/(cyc*1024):.3f}  "
          f"LDS bank-conflict cycles/LDS inst={g('SQ_LDS_BANK_CONFLICT')/max(g('SQ_INSTS_LDS'),1):.2f}  wave occupancy (SQ_WAVE_CYCLES/(cycles*1024))={g('SQ_WAVE_CYCLES')/(cyc*1024):.2f}")
    print(f"   HBM read/dispatch = 2*FETCH_SIZE*1024 = {2*g('FETCH_SIZE')*1024:.3e} B, write = WRITE_SIZE*1024 = {g('WRITE_SIZE')*1024:.3e} B")
PY
cat $OUT/summary.txt
# keep what the summaries cite, drop the bulky raw traces (gpurun merges at most 64 MiB back)
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*agent_info.csv" -delete
