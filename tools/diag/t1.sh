timeout -k 10 600 python -m pytest tests/test_gpu_encoder.py -x -q -k "negative" 2>&1 | tail -12
