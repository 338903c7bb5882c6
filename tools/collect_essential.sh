#!/bin/bash
# The part of tools/collect_profiles.sh the bench line and the roofline claims depend on (bench, kernel stats, FETCH / WRITE / MFMA
# counter passes, per-shape GEMM bench, small-batch table, decode bench):  tools/collect_essential.sh <tag> -> gpurun_out/<tag>/
TAG=${1:-latest}
R=$(pwd); OUT=$R/gpurun_out/$TAG; RAW=/tmp/prof_$TAG
mkdir -p $OUT $RAW
python bench.py 2>/dev/null | tail -1 > $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-search-leg --no-small-batch --no-config5-line --no-stream-leg"
rocprofv3 --kernel-trace --stats -d $RAW/stats -o k -- $B --steps 3 --warmup 1 > /dev/null 2>&1
for c in "FETCH_SIZE:f" "WRITE_SIZE:w" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES:m"; do
  n=${c##*:}; ctr=${c%%:*}
  rocprofv3 --pmc $ctr --kernel-trace -d $RAW/pmc_$n -o pmc -- $B --steps 1 --warmup 0 > /dev/null 2>&1 || echo "pass $n failed"
done
cd $R
python tools/rocpd_summary.py $RAW/stats/k_results.db > $OUT/kernel_stats.csv
python tools/pmc_summary.py $RAW/pmc_f/pmc_results.db $RAW/pmc_w/pmc_results.db $RAW/pmc_m/pmc_results.db > $OUT/pmc.json
python tools/gemm_bench.py --iters 40 > $OUT/gemm_bench.txt 2>/dev/null
for b in 1 2 4 8; do
  python bench.py --batch $b --steps 20 --warmup 3 --no-cpu-baseline --no-search-leg --no-small-batch --no-config5-line --no-stream-leg 2>/dev/null | tail -1 > $OUT/batch_$b.json
done
python tools/vqa_bench.py --out $OUT/vqa_bench.json > /dev/null 2>&1 || true
ls -la $OUT
