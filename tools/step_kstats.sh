# usage (GPU box): bash tools/step_kstats.sh <tag> [bench_train args]  -- per-kernel time of the whole airplane train step (eager, 5 + 2 steps)
TAG=${1:-x}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/kstats_${TAG}; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/full -o full -- python3 tools/bench_train.py --steps 5 "$@" > $OUT/full.log 2>&1
python3 - > $OUT/summary.txt 2>&1 <<PY
import csv, glob
ff = glob.glob("$OUT/full/*kernel_stats.csv")
fr = [r for r in csv.DictReader(open(ff[0])) if 'copyBuffer' not in r['Name']]
tot = sum(float(r['TotalDurationNs']) for r in fr) / 7 / 1e3
print(f"kernel time per eager step: {tot:.0f} us, kernels per step: {sum(int(r['Calls']) for r in fr)/7:.0f}")
for r in fr[:70]:
    print(f"{float(r['TotalDurationNs'])/7/1e3:9.1f} us/step  calls/step={int(r['Calls'])/7:7.1f} avg_us={float(r['AverageNs'])/1e3:9.2f}  {r['Name'][:110]}")
PY
cat $OUT/summary.txt
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
