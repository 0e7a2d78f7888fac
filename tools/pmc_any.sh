# usage (GPU box): bash tools/pmc_any.sh <tag> '<regex of kernel names>' <python script and args...>
# rocprofv3 kernel-trace stats + PMC passes (each in its own run) of any python command; per kernel: duration, instructions per wave,
# matrix-pipe busy, VALU issue share, waits, LDS conflicts, HBM bytes.  Summary -> gpurun_out/pmc_<tag>/summary.txt
TAG=$1; RE=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_${TAG}; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ts -o ts -- python3 "$@" > $OUT/ts.log 2>&1
timeout -k 10 150 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc1 -o ts -- python3 "$@" > $OUT/pmc1.log 2>&1
timeout -k 10 150 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/pmc2 -o ts -- python3 "$@" > $OUT/pmc2.log 2>&1
timeout -k 10 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -o ts -- python3 "$@" > $OUT/pmc3.log 2>&1
timeout -k 10 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc4 -o ts -- python3 "$@" > $OUT/pmc4.log 2>&1
python3 - "$RE" > $OUT/summary.txt 2>&1 <<PY
import csv, glob, collections, re, sys
out, rx = "$OUT", re.compile(sys.argv[1])
short = lambda n: re.sub(r"\(anonymous namespace\)::|^void ", "", n).split("(")[0][:60]
dur = {}
for r in csv.DictReader(open(glob.glob(f"{out}/ts/*kernel_stats.csv")[0])):
    if rx.search(r["Name"]): dur[short(r["Name"])] = (float(r["AverageNs"]) / 1e3, int(r["Calls"]))
agg = collections.defaultdict(lambda: collections.defaultdict(list)); meta = {}
for d in ("pmc1", "pmc2", "pmc3", "pmc4"):
    for fn in glob.glob(f"{out}/{d}/*counter_collection.csv"):
        for r in csv.DictReader(open(fn)):
            if not rx.search(r["Kernel_Name"]): continue
            key = short(r["Kernel_Name"])
            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[key] = {m: r[m] for m in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "Accum_VGPR_Count", "Scratch_Size")}
for key in sorted(agg, key=lambda k: -dur.get(k, (0, 0))[0] * dur.get(k, (0, 0))[1]):
    a = agg[key]; g = lambda c: sum(a[c]) / len(a[c]) if a.get(c) else float("nan")
    waves = float(meta[key]["Grid_Size"]) / 64; cyc = g("GRBM_GUI_ACTIVE") / 8
    print(f"-- {key}: avg {dur.get(key, (float('nan'), 0))[0]:.1f} us x {dur.get(key, (0, 0))[1]} calls  {meta[key]}")
    print(f"   waves={waves:.0f} per wave: VALU={g('SQ_INSTS_VALU')/waves:.0f} MFMA={g('SQ_INSTS_MFMA')/waves:.0f} LDS={g('SQ_INSTS_LDS')/waves:.0f} SALU={g('SQ_INSTS_SALU')/waves:.0f} "
          f"VMEM_RD={g('SQ_INSTS_VMEM_RD')/waves:.0f} VMEM_WR={g('SQ_INSTS_VMEM_WR')/waves:.0f}")
    print(f"   cycles~{cyc:.3e}  MFMA pipe busy={g('SQ_VALU_MFMA_BUSY_CYCLES')/(cyc*1024):.3f}  VALU issue={4*g('SQ_INSTS_VALU')/(cyc*1024):.3f}  "
          f"wave occupancy={g('SQ_WAVE_CYCLES')/(cyc*1024):.2f} waves/SIMD  wait_any/wave_cycles={g('SQ_WAIT_ANY')/max(g('SQ_WAVE_CYCLES'),1):.2f}  "
          f"wait_inst/wave_cycles={g('SQ_WAIT_INST_ANY')/max(g('SQ_WAVE_CYCLES'),1):.2f}  LDS conflict cycles/inst={g('SQ_LDS_BANK_CONFLICT')/max(g('SQ_INSTS_LDS'),1):.2f}")
    print(f"   HBM read = 2*FETCH_SIZE*1024 = {2*g('FETCH_SIZE')*1024/1e6:.1f} MB, write = WRITE_SIZE*1024 = {g('WRITE_SIZE')*1024/1e6:.1f} MB per dispatch")
PY
cat $OUT/summary.txt
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*agent_info.csv" -delete
