# usage (GPU box): bash docs/experiments/probes/train_kernel_times.sh [extra bench_train args] -- per-kernel average durations of the whole airplane train step (eager)
export TMPDIR=/tmp
OUT=gpurun_out/prof_tk; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 tools/bench_train.py --steps 5 "$@" > $OUT/run.log 2>&1
f=$(find $OUT -name "*kernel_stats.csv" | tail -1)
test -n "$f" && python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'copyBuffer' not in r['Name'] and 'FillFunctor' not in r['Name']]
print(f"kernel time per eager step: {sum(float(r['TotalDurationNs']) for r in rows)/7/1e6:.2f} ms")
for r in rows[:24]:
    print(f"{r['Name'][:96]:96s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:9.2f}")
PY
find $OUT -name "*kernel_trace.csv" -delete
