# usage (GPU box): bash tools/enc_kstats.sh  -- per-kernel durations of the encoder train pipeline inside the airplane train step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pf
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pf -o full -- python3 tools/bench_train.py --steps 5 "$@" > gpurun_out/pf.log 2>&1
python3 - <<PY
import csv, glob
rows = list(csv.DictReader(open(glob.glob("gpurun_out/pf/*kernel_stats.csv")[0])))
tot = 0
for r in rows:
    if "enc_" in r["Name"]:
        t = float(r["TotalDurationNs"]) / 7e3; tot += t
        print(r["Name"].replace("(anonymous namespace)::", "")[:70].ljust(70), r["Calls"].rjust(4), "avg_us %8.1f  us/step %8.1f" % (float(r["AverageNs"]) / 1e3, t))
print("encoder train kernels, us/step:", round(tot, 1))
PY
rm -rf gpurun_out/pf
