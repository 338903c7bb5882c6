#!/bin/bash
# Re-collects the evidence under profiles/ on a GPU box (run from the repo root, e.g. through gpurun):
#   tools/collect_profiles.sh <tag>   ->  gpurun_out/<tag>/{bench.json, kernel_stats.csv, pmc.json, sq_counters.json, bench_fp8.json, ...}
# Raw rocprofv3 databases stay in /tmp on the box (gpurun_out is limited to 64 MiB); only the summaries are kept.
# Counter passes are separate rocprofv3 runs with --kernel-trace only (one counter set per pass), as MI355X_MICROARCH.md prescribes.
set -e
TAG=${1:-latest}
R=$(pwd)
OUT=$R/gpurun_out/$TAG
RAW=/tmp/prof_$TAG
mkdir -p $OUT $RAW
python bench.py 2>/dev/null | tail -1 > $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-search-leg --no-small-batch --no-config5-line"
rocprofv3 --kernel-trace --stats -d $RAW/stats -o k -- $B --steps 3 --warmup 1 > /dev/null 2>&1
for c in "FETCH_SIZE:f" "WRITE_SIZE:w" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES:m" \
         "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA:s1" \
         "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU:s2" \
         "TCC_HIT_sum TCC_MISS_sum:t"; do
  n=${c##*:}; ctr=${c%%:*}
  rocprofv3 --pmc $ctr --kernel-trace -d $RAW/pmc_$n -o pmc -- $B --steps 1 --warmup 0 > /dev/null 2>&1 || echo "pass $n failed"
done
cd $R
python tools/rocpd_summary.py $RAW/stats/k_results.db > $OUT/kernel_stats.csv
python tools/pmc_summary.py $RAW/pmc_f/pmc_results.db $RAW/pmc_w/pmc_results.db $RAW/pmc_m/pmc_results.db > $OUT/pmc.json
python tools/pmc_dump.py $RAW/pmc_s1/pmc_results.db $RAW/pmc_s2/pmc_results.db $RAW/pmc_t/pmc_results.db > $OUT/sq_counters.json 2>/dev/null || true
python bench.py --fp8 --no-cpu-baseline --no-small-batch 2>/dev/null | tail -1 > $OUT/bench_fp8.json
python tools/gemm_bench.py --iters 40 > $OUT/gemm_bench.txt 2>/dev/null
python tools/attn_kernel_bench.py > $OUT/attn_kernel_bench.txt 2>/dev/null
python tools/vqa_bench.py --out $OUT/vqa_bench.json > /dev/null 2>&1 || true
python tools/cue_bench.py 2>/dev/null | tail -1 > $OUT/cue_bench.json || true
ls -la $OUT
