export GWTF_FORCE_SHARDED=1
bash tools/step_kstats.sh sh | grep -E "us/step|kernel time" | grep -v "bwd_kernel\|stack_kernel\|stats_kernel\|enc_\|FillFunctor\|tail\|hidden_kernel\|hid_bwd\|kept_bwd\|out_kernel\|out_bwd\|head_" | head -42 | cut -c1-170
