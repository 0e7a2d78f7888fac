# usage: bash tools/sweep.sh  -- GPU tests then bench sweeps over the tuning hook; prints a compact table
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; tail -2 gpurun_out/pytest_gpu.log
rm -f gpurun_out/sweep.log
for w in m1 airplane; do for ppw in 0 16 32 64; do
  echo "## $w ppw=$ppw" >> gpurun_out/sweep.log
  timeout -k 10 120 python bench.py --workload $w --points-per-wave $ppw --no-cpu-baseline >> gpurun_out/sweep.log 2>&1
done; done
for w in ae k16; do echo "## $w" >> gpurun_out/sweep.log; timeout -k 10 120 python bench.py --workload $w --no-cpu-baseline >> gpurun_out/sweep.log 2>&1; done
python - <<PY
import json
tag=""
for l in open("gpurun_out/sweep.log"):
    if l.startswith("##"): tag=l.strip(); continue
    try: d=json.loads(l)
    except Exception:
        if "Error" in l or "error" in l: print(l.strip()[:200])
        continue
    print(tag, d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"])
PY
