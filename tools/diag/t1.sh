timeout -k 10 600 python -m pytest tests/test_gpu_heads.py tests/test_gpu_models.py -x -q 2>&1 | tail -3
bash tools/step_kstats.sh k6 2>&1 | grep -E "head_"
for i in 1 2; do timeout -k 10 200 python tools/bench_train.py --graph --steps 30 2>&1 | grep -E "hipGraph"; done
