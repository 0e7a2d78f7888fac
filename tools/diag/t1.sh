timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "full_tile_sizes" 2>&1 | tail -3
