# usage (GPU box): bash docs/experiments/probes/graph_busy.sh -- busy fraction of the GPU inside one replay of the graphed training step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/gb; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT -o gb -- python3 tools/bench_train.py --graph --steps 6 > $OUT/run.log 2>&1
tail -2 $OUT/run.log
python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open("$OUT/gb_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
end = int(rows[-1]["End_Timestamp"])
win = [r for r in rows if int(r["Start_Timestamp"]) >= end - 60_000_000]     # the last 60 ms: two replays
t0 = int(win[0]["Start_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in win)
span = end - t0
gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(win, win[1:])]
pos = [g for g in gaps if g > 0]
print(f"window {span/1e6:.2f} ms, {len(win)} kernels, busy {busy/1e6:.2f} ms ({100*busy/span:.1f} %), "
      f"gaps: {len(pos)} positive, total {sum(pos)/1e6:.2f} ms, mean {sum(pos)/max(1,len(pos))/1e3:.2f} us, overlapped starts {len(gaps)-len(pos)}")
fam = collections.Counter()
for r in win:
    n = r["Kernel_Name"]; d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    key = next((k for k in ("bwd_kernel", "stack_kernel", "stats_kernel", "fold", "combine", "dw1", "adam", "film", "nll", "encoder", "Cijk", "MIOpen", "elementwise", "reduce", "Cat", "copyBuffer", "fill") if k.lower() in n.lower()), "other")
    fam[key] += d
for k, v in fam.most_common(): print(f"  {k:14s} {v/1e6/ (span/26.7e6):8.2f} ms per 26.7 ms step")
top = collections.defaultdict(lambda: [0, 0])
for r in win:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); top[r["Kernel_Name"]][0] += d; top[r["Kernel_Name"]][1] += 1
for n, (d, c) in sorted(top.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"  {d/1e6/(span/26.7e6):7.2f} ms/step  {c/(span/26.7e6):7.1f} calls/step  avg {d/c/1e3:8.1f} us  {n[:110]}")
PY
rm -rf $OUT
