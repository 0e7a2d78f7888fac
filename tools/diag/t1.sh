mkdir -p gpurun_out/t1
timeout -k 10 1100 python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee gpurun_out/t1/pytest.txt
