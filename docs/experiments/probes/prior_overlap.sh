# usage (GPU box): bash docs/experiments/probes/prior_overlap.sh -- where do the prior-flow kernels sit on the timeline of a graphed training step?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/po; mkdir -p gpurun_out/po
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/po -o t -- python3 tools/bench_train.py --graph --steps 6 --batch ${1:-64} > gpurun_out/po.log 2>&1
python3 - <<PY
import csv, glob
rows = list(csv.DictReader(open(glob.glob("gpurun_out/po/*kernel_trace.csv")[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last occurrence of prior_bwd = inside the last replay; take the step around it
idx = max(i for i, r in enumerate(rows) if "prior_bwd" in r["Kernel_Name"])
pb = rows[idx]; t1 = int(pb["End_Timestamp"])
pf = max((r for r in rows[:idx] if "prior_fwd" in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
t0 = int(pf["Start_Timestamp"])
step = [r for r in rows if int(r["Start_Timestamp"]) >= t0 - 2_000_000 and int(r["Start_Timestamp"]) <= t1 + 3_000_000]
def us(t): return (int(t) - t0) / 1e3
print("prior_fwd %.0f .. %.0f us   prior_bwd %.0f .. %.0f us" % (us(pf["Start_Timestamp"]), us(pf["End_Timestamp"]), us(pb["Start_Timestamp"]), us(pb["End_Timestamp"])))
def first_last(name):
    k = [r for r in step if name in r["Kernel_Name"]]
    return (us(k[0]["Start_Timestamp"]), us(k[-1]["End_Timestamp"]), len(k)) if k else None
for n in ("enc_train_fwd", "stats_kernel", "stack_kernel", "nll", "bwd_kernel", "bwd_tail2", "enc_train_bwd", "adam"):
    print(n, first_last(n))
# gaps on the non-prior timeline larger than 50 us
others = [r for r in step if "prior_" not in r["Kernel_Name"]]
prev_end = None
for r in others:
    s = int(r["Start_Timestamp"])
    if prev_end is not None and s - prev_end > 50_000:
        print("gap %.0f us before %s at %.0f us" % ((s - prev_end) / 1e3, r["Kernel_Name"][:60], us(s)))
    prev_end = max(prev_end or 0, int(r["End_Timestamp"]))
PY
rm -rf gpurun_out/po
