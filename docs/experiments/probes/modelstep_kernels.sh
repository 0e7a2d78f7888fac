# usage (GPU box): bash docs/experiments/probes/modelstep_kernels.sh  -- kernel-family breakdown of the graphed end-to-end training step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/ms; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ms -- python3 tools/bench_train.py --graph --steps 10 > $OUT/run.log 2>&1
tail -3 $OUT/run.log
python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open("$OUT/ms_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot / 1e6)
for r in rows[:28]:
    print(r["Name"][:90].ljust(90), r["Calls"].rjust(7), "avg_us %8.1f" % (float(r["AverageNs"]) / 1e3), "tot_ms %8.2f" % (float(r["TotalDurationNs"]) / 1e6))
PY
