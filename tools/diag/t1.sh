timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py tests/test_gpu_models.py tests/test_gpu_fullgrid.py tests/test_gpu_robustness.py tests/test_gpu_advice.py -x -q 2>&1 | tail -3
for i in 1 2; do timeout -k 10 200 python tools/bench_train.py --graph --steps 30 2>&1 | grep -E "hipGraph"; done
