mkdir -p gpurun_out/t1
timeout -k 10 900 python -m pytest tests/test_gpu_heads.py tests/test_gpu_prior.py tests/test_gpu_film_heads.py -q 2>&1 | tail -15 > gpurun_out/t1/pytest_rows.txt
cat gpurun_out/t1/pytest_rows.txt
