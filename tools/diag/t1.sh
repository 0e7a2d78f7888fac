for i in 1 2; do timeout -k 10 200 python tools/bench_train.py --graph --steps 30 2>&1 | grep -E "hipGraph"; done
bash tools/step_kstats.sh t7 | grep -E "bwd_kernel|kernel time"
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "train or graph or fused or sgd or directional or reproducible or backward" 2>&1 | tail -3
