for i in 1 2; do timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q 2>&1 | grep -E "^E  +(Assert|assert)|passed|failed|^FAILED" | head -6; done
