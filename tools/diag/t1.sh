timeout -k 10 600 python -m pytest tests/test_gpu_encoder.py -x -q 2>&1 | tail -2
bash tools/enc_kstats.sh --parts e 2>&1 | grep -E "dw_kernel|dw_reduce"
