timeout -k 10 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_parity.py tests/test_gpu_advice.py -x -q -k "rank or graph or dist or parallel or overlap or accum or sync" 2>&1 | tail -4
for i in 1 2; do GWTF_FORCE_SHARDED=1 timeout -k 10 200 python tools/bench_train.py --graph --steps 30 2>&1 | grep -E "hipGraph"; done
timeout -k 10 200 python tools/bench_train.py --graph --steps 30 2>&1 | grep -E "hipGraph"
