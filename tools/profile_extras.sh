# usage (GPU box): bash tools/profile_extras.sh <tag>  -- kernel-trace stats (+ PMC for the encoder kernel) of the rows built
# after the headline path: PointNet encoder, structural losses, the train step.  Summary -> gpurun_out/prof_<tag>_extras/summary.txt
TAG=${1:-r05}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_${TAG}_extras; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/enc -o enc -- python3 tools/bench_encoder.py 64 > $OUT/enc.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/enc_pmc -o enc -- python3 tools/bench_encoder.py 64 > $OUT/enc_pmc.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/enc_fetch -o enc -- python3 tools/bench_encoder.py 64 > $OUT/enc_fetch.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/met -o met -- python3 tools/bench_metrics.py 64 > $OUT/met.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ts -o ts -- python3 tools/diag/trainstep_kernels.py > $OUT/ts.log 2>&1
python3 - > $OUT/summary.txt 2>&1 <<PY
import csv, glob, collections
out = "$OUT"
def stats(d, pat, n=8):
    f = glob.glob(f"{out}/{d}/*kernel_stats.csv")
    if not f: print("  (no stats)"); return
    for i, r in enumerate(csv.DictReader(open(f[0]))):
        if i < n or any(p in r["Name"] for p in pat):
            print(f"  {r['Name'][:86]:86s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:10.2f} pct={r['Percentage']}")
print("===== PointNet encoder, 64 x 2048 points (tools/bench_encoder.py) =====")
print("\n".join("  " + l for l in open(f"{out}/enc.log").read().splitlines() if " ms" in l and "rocprof" not in l))
stats("enc", ["encoder_kernel"], 4)
agg = collections.defaultdict(list); meta = {}
for d in ("enc_pmc", "enc_fetch"):
    for f in glob.glob(f"{out}/{d}/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "encoder_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "Accum_VGPR_Count")}
print("-- PMC, encoder_kernel, mean per dispatch --", meta)
for k in sorted(agg): print(f"  {k:28s} {sum(agg[k])/len(agg[k]):16.1f}  (n={len(agg[k])})")
if agg.get("SQ_INSTS_MFMA"):
    waves = float(meta["Grid_Size"]) / 64
    print(f"  waves={waves:.0f} VALU/wave={sum(agg['SQ_INSTS_VALU'])/len(agg['SQ_INSTS_VALU'])/waves:.0f} MFMA/wave={sum(agg['SQ_INSTS_MFMA'])/len(agg['SQ_INSTS_MFMA'])/waves:.0f}")
if agg.get("FETCH_SIZE"):
    print(f"  HBM read per dispatch = 2*FETCH_SIZE*1024 = {2*sum(agg['FETCH_SIZE'])/len(agg['FETCH_SIZE'])*1024:.3e} B (gfx950 correction x2)")
print("===== structural losses, 64 pairs of 2048-point clouds (tools/bench_metrics.py) =====")
print("\n".join("  " + l for l in open(f"{out}/met.log").read().splitlines() if " ms" in l and "rocprof" not in l))
stats("met", ["nnd_kernel", "emd_"], 8)
print("===== train step, one 33-coupling component, 64 x 2048 (tools/diag/trainstep_kernels.py, eager, 7 steps) =====")
f = glob.glob(f"{out}/ts/*kernel_stats.csv")
if f:
    rows = list(csv.DictReader(open(f[0])))
    print(f"  kernel time per step: {sum(float(r['TotalDurationNs']) for r in rows)/7/1e6:.2f} ms, kernels per step: {sum(int(r['Calls']) for r in rows)/7:.0f}")
stats("ts", [], 14)
PY
cat $OUT/summary.txt
