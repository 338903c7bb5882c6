#!/bin/bash
# Re-collects the evidence under profiles/ on a GPU box (run from the repo root, e.g. through gpurun):
#   tools/collect_profiles.sh <tag>            ->  gpurun_out/<tag>/{bench.json, kernel_stats.csv, pmc.json, bench_fp8.json, ...}
# Counter passes are separate rocprofv3 runs with --kernel-trace only (one counter set per pass), as MI355X_MICROARCH.md prescribes.
set -e
TAG=${1:-latest}
R=$(pwd)
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
python bench.py 2>/dev/null | tail -1 > $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o k -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
for c in "FETCH_SIZE:f" "WRITE_SIZE:w" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES:m"; do
  n=${c##*:}; ctr=${c%%:*}
  rocprofv3 --pmc $ctr --kernel-trace -d $OUT/pmc_$n -o pmc -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
done
cd $R
python tools/rocpd_summary.py $OUT/stats/k_results.db > $OUT/kernel_stats.csv
python tools/pmc_summary.py $OUT/pmc_f/pmc_results.db $OUT/pmc_w/pmc_results.db $OUT/pmc_m/pmc_results.db > $OUT/pmc.json
python bench.py --fp8 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_fp8.json
python tools/vqa_bench.py --out $OUT/vqa_bench.json > /dev/null 2>&1
python tools/search_bench.py --targets 8 --device-reductions 2>/dev/null | tail -1 > $OUT/search_bench.json
python tools/cue_bench.py 2>/dev/null | tail -1 > $OUT/cue_bench.json
ls -la $OUT
