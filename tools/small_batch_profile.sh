#!/bin/bash
# Latency regime (what each rank of an 8-way sharded search sees): bench.py --batch 1/2/4/8 lines + rocprofv3 kernel stats at B=1,4.
#   tools/small_batch_profile.sh <tag>  ->  gpurun_out/<tag>/{batch_B.json, kernel_stats_bB.csv}
TAG=${1:-smallb}
R=$(pwd); OUT=$R/gpurun_out/$TAG; RAW=/tmp/prof_$TAG
mkdir -p $OUT $RAW
for b in ${BATCHES:-1 2 4 8}; do
  python bench.py --batch $b --steps 20 --warmup 3 --no-cpu-baseline --no-search-leg --no-small-batch --no-config5-line 2>/dev/null | tail -1 > $OUT/batch_$b.json
done
cd /tmp && export TMPDIR=/tmp
for b in ${PROF_BATCHES:-1 4}; do
  rocprofv3 --kernel-trace --stats -d $RAW/stats_$b -o k -- python $R/bench.py --batch $b --steps 6 --warmup 1 --no-cpu-baseline --no-search-leg --no-small-batch --no-config5-line > /dev/null 2>&1
  python $R/tools/rocpd_summary.py $RAW/stats_$b/k_results.db > $OUT/kernel_stats_b$b.csv
done
cd $R; ls -la $OUT
