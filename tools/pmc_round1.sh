cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 -L > gpurun_out/counters_list.txt 2>&1
timeout -k 10 300 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; tail -2 gpurun_out/pytest_gpu.log
for w in m1 airplane; do timeout -k 10 120 python bench.py --workload $w --no-cpu-baseline >> gpurun_out/bench2.log 2>&1; done
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d gpurun_out/pmc1 -o m1 -- python3 bench.py --workload m1 --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/pmc1.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_SALU --output-format csv -d gpurun_out/pmc2 -o m1 -- python3 bench.py --workload m1 --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/pmc2.log 2>&1
tail -3 gpurun_out/pmc1.log gpurun_out/pmc2.log
python - <<PY
import json
for l in open("gpurun_out/bench2.log"):
    try: d=json.loads(l)
    except Exception: continue
    print(d["config"]["workload"][:20], d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"])
PY
