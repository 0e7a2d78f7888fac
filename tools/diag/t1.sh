for lib in build_ab/libgwtf_dwplain.so go_with_the_flows_amd/libgwtf_hip.so; do echo "== $lib"; for i in 1 2; do timeout -k 10 200 python tools/bench_train.py --graph --steps 30 --lib $lib 2>&1 | grep -E "hipGraph"; done; done
bash tools/step_kstats.sh k8 --lib build_ab/libgwtf_dwplain.so 2>&1 | grep -E "bwd_kernel<3, 2, 3|bwd_tail1"
