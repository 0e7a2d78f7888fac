# usage (GPU box): bash tools/step_kcount.sh  -- kernels of one airplane training step by number of launches (eager, rocprofv3)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pf
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pf -o full -- python3 tools/bench_train.py --steps 5 > gpurun_out/pf.log 2>&1
python3 - <<PY
import csv, glob
rows = [r for r in csv.DictReader(open(glob.glob("gpurun_out/pf/*kernel_stats.csv")[0])) if "copyBuffer" not in r["Name"]]
tot_calls = sum(int(r["Calls"]) for r in rows) / 7; tot_t = sum(float(r["TotalDurationNs"]) for r in rows) / 7e6
print("kernels/step %.0f   kernel ms/step %.2f" % (tot_calls, tot_t))
ours = [r for r in rows if "anonymous namespace" in r["Name"] and "at::" not in r["Name"]]
print("hand-written: launches/step %.0f  ms/step %.2f" % (sum(int(r["Calls"]) for r in ours) / 7, sum(float(r["TotalDurationNs"]) for r in ours) / 7e6))
rows.sort(key=lambda r: -int(r["Calls"]))
for r in rows[:40]:
    print("%6.1f calls/step  avg %7.1f us  %8.1f us/step  %s" % (int(r["Calls"]) / 7, float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 7e3, r["Name"][:110]))
PY
rm -rf gpurun_out/pf
