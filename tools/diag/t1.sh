mkdir -p gpurun_out/t1
timeout -k 10 1100 python -m pytest tests -m gpu -q 2>&1 | tail -3 | tee gpurun_out/t1/pytest.txt
timeout -k 10 600 python bench.py > gpurun_out/t1/bench.json 2> gpurun_out/t1/bench.err; tail -c 200 gpurun_out/t1/bench.json
