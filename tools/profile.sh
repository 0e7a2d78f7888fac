# usage (GPU box): bash tools/profile.sh <tag>   -- kernel-trace stats + PMC passes for the bench workloads
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_$TAG; mkdir -p $OUT
for w in airplane m1; do
  # the SAME command as the bench (one hipGraph per step): its own kernel_ms (HIP events) and rocprofv3's average must agree
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$w -o $w -- python3 bench.py --workload $w --no-cpu-baseline --no-also --steps 200 > $OUT/trace_$w.log 2>&1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $OUT/pmc_sq_$w -o $w -- python3 bench.py --workload $w --no-cpu-baseline --no-also --eager --steps 5 --warmup 2 > $OUT/pmc_sq_$w.log 2>&1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma_$w -o $w -- python3 bench.py --workload $w --no-cpu-baseline --no-also --eager --steps 5 --warmup 2 > $OUT/pmc_mfma_$w.log 2>&1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$w -o $w -- python3 bench.py --workload $w --no-cpu-baseline --no-also --eager --steps 5 --warmup 2 > $OUT/pmc_fetch_$w.log 2>&1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$w -o $w -- python3 bench.py --workload $w --no-cpu-baseline --no-also --eager --steps 5 --warmup 2 > $OUT/pmc_write_$w.log 2>&1
done
# HBM traffic of the stack kernel for the other bench shapes (roofline.traffic of the `also` records): FETCH_SIZE / WRITE_SIZE only
for w in ae svr k16; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$w -o $w -- python3 bench.py --workload $w --no-cpu-baseline --no-also --eager --steps 5 --warmup 2 > $OUT/pmc_fetch_$w.log 2>&1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$w -o $w -- python3 bench.py --workload $w --no-cpu-baseline --no-also --eager --steps 5 --warmup 2 > $OUT/pmc_write_$w.log 2>&1
done
python3 tools/summarize_profile.py $OUT $TAG > $OUT/summary.txt 2>&1
for w in airplane m1; do echo "bench.py's own line under the profiler ($w): $(grep -o '"value": [0-9.]*' $OUT/trace_$w.log | head -1), $(grep -o '"kernel_ms": [0-9.]*' $OUT/trace_$w.log | head -1)" >> $OUT/summary.txt; done
cat $OUT/summary.txt
# keep what the summaries cite, drop the bulky raw traces (gpurun merges at most 64 MiB back)
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*agent_info.csv" -delete
