export TMPDIR=/tmp
OUT=gpurun_out/prof_ft; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 tools/diag/film_train_time.py > $OUT/run.log 2>&1
f=$(find $OUT -name "*kernel_stats.csv" | tail -1)
test -n "$f" && python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
n = 114.0   # 3 warm-up + capture + 110 replays
for r in rows[:25]:
    print(f"{float(r['TotalDurationNs'])/n/1e3:7.1f} us/step calls/step={int(r['Calls'])/n:5.1f} avg={float(r['AverageNs'])/1e3:7.1f}  {r['Name'][:120]}")
PY
find $OUT -name "*kernel_trace.csv" -delete
