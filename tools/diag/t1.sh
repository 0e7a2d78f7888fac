for d in 1 2 3; do echo "== DBG $d"; bash tools/enc_kstats.sh --parts e --lib build_ab/libgwtf_encdbg$d.so 2>&1 | grep -E "bwd_kernel"; done
echo "== base"; bash tools/enc_kstats.sh --parts e 2>&1 | grep -E "bwd_kernel"
