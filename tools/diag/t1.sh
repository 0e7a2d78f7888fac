mkdir -p gpurun_out/t1
bash tools/profile.sh r15 > gpurun_out/prof_r15.log 2>&1
bash tools/profile_train.sh r15 > gpurun_out/prof_r15_train.log 2>&1
bash tools/pmc_any.sh enc15 'enc_' tools/bench_train.py --parts e --steps 4 > /dev/null 2>&1
timeout -k 10 600 python bench.py > gpurun_out/t1/bench.json 2> gpurun_out/t1/bench.err
tail -c 300 gpurun_out/t1/bench.json
