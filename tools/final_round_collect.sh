set -x
R=$(pwd); OUT=$R/gpurun_out/r03c; mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $OUT/gpu_tests.txt
tools/collect_essential.sh r03c > /dev/null 2>&1
python bench.py --fp8 --no-cpu-baseline --no-small-batch 2>/dev/null | tail -1 > $OUT/bench_fp8.json
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-search-leg --no-small-batch --no-config5-line --no-stream-leg"
RAW=/tmp/prof_r03c
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA:s1" \
         "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU:s2" \
         "TCC_HIT_sum TCC_MISS_sum:t"; do
  n=${c##*:}; ctr=${c%%:*}
  rocprofv3 --pmc $ctr --kernel-trace -d $RAW/pmc_$n -o pmc -- $B --steps 1 --warmup 0 > /dev/null 2>&1 || echo "pass $n failed"
done
cd $R
python tools/pmc_dump.py $RAW/pmc_s1/pmc_results.db $RAW/pmc_s2/pmc_results.db $RAW/pmc_t/pmc_results.db > $OUT/sq_counters.json 2>/dev/null || true
cat $OUT/gpu_tests.txt; ls -la $OUT
