timeout -k 10 600 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_models.py -x -q 2>&1 | tail -3
bash tools/enc_kstats.sh 2>&1 | grep -E "fwd_kernel|bwd_kernel|dw_kernel|us/step:"
for i in 1 2; do timeout -k 10 200 python tools/bench_train.py --graph --steps 30 2>&1 | grep -E "hipGraph"; done
