mkdir -p gpurun_out/t1
( timeout -k 10 200 python tools/bench_train.py --graph --steps 30 2>&1 | grep -E "hipGraph|eager"
  GWTF_FORCE_SHARDED=1 timeout -k 10 200 python tools/bench_train.py --graph --steps 30 2>&1 | grep -E "hipGraph|eager|all-reduces" ) | tee gpurun_out/t1/train.txt
bash tools/step_kstats.sh t6 | grep -E "film_heads|Cijk|pack_|kernel time"
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "train or graph or fused or sgd or syncbn or list_slot or directional or film_heads or encoder" 2>&1 | tail -3 | tee gpurun_out/t1/pytest.txt
