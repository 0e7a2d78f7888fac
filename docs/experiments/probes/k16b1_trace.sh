# usage (GPU box): bash docs/experiments/probes/k16b1_trace.sh -- per-kernel durations of the one-shape K=16 sampling call (configs[3], reference batch of one)
export TMPDIR=/tmp
OUT=gpurun_out/prof_k16b1; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 bench.py --workload k16_b1 --no-cpu-baseline --no-also --steps 200 > $OUT/run.log 2>&1
f=$(find $OUT -name "*kernel_stats.csv" | tail -1)
test -n "$f" && python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f"{r['Name'][:100]:100s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:8.2f} min_us={float(r['MinNs'])/1e3:8.2f}")
PY
tail -1 $OUT/run.log | cut -c1-300
