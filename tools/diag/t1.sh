for v in nb2 nbt1; do echo "== $v"; bash tools/enc_kstats.sh --parts e --lib build_ab/libgwtf_$v.so 2>&1 | grep -E "bwd_kernel"; done
echo "== base"; bash tools/enc_kstats.sh --parts e 2>&1 | grep -E "bwd_kernel"
