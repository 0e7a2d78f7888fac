timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py tests/test_gpu_models.py -x -q -k "train or grad or backward or fused or sgd or step or reproduc" 2>&1 | tail -3
bash tools/step_kstats.sh k9 2>&1 | grep -E "bwd_kernel<3"
for i in 1 2; do timeout -k 10 200 python tools/bench_train.py --graph --steps 30 2>&1 | grep -E "hipGraph"; done
