bash tools/pmc_any.sh k4step 'bwd_kernel|stack_kernel|stats_kernel|bwd_tail1|fold' tools/bench_train.py --steps 3 > /dev/null 2>&1
head -40 gpurun_out/pmc_k4step/summary.txt
