# usage (GPU box): bash tools/kstats.sh [workloads...]  -- per-kernel average durations under rocprofv3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for w in ${@:-m1 airplane}; do
  rm -rf gpurun_out/ks_$w
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks_$w -o $w -- python3 bench.py --workload $w --no-cpu-baseline --no-live-traffic --steps 50 > gpurun_out/ks_$w.log 2>&1
  python3 - <<PY
import csv, json
for r in csv.DictReader(open("gpurun_out/ks_$w/${w}_kernel_stats.csv")):
    if any(k in r["Name"] for k in ("stack_kernel", "film", "nll", "pack_")):
        print("$w", r["Name"][:58].ljust(58), "calls", r["Calls"].rjust(5), "avg_us %8.2f" % (float(r["AverageNs"]) / 1e3))
for l in open("gpurun_out/ks_$w.log"):
    if l.startswith("{"):
        d = json.loads(l); print("$w step_ms", d["ms_per_step"], "value", d["value"])
PY
done
