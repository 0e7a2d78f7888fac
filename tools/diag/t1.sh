for v in encdbg4 dwmap0 dwmap1; do echo "== $v"; bash tools/enc_kstats.sh --parts e --lib build_ab/libgwtf_$v.so 2>&1 | grep -E "dw_kernel"; done
