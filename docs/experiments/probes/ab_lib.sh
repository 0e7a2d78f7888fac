for i in 1 2 3; do
for lib in go_with_the_flows_amd/libgwtf_base.so go_with_the_flows_amd/libgwtf_hip.so; do
  echo -n "$lib: "; timeout -k 10 100 python bench.py --workload airplane --no-cpu-baseline --no-also --steps 300 --lib $lib 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['kernel_ms'])"
done; done
