#!/bin/bash
# Round-4 final evidence: tools/collect_r04.sh  ->  gpurun_out/r04f/  (summaries only; raw rocprofv3 databases stay in /tmp)
R=$(pwd); OUT=$R/gpurun_out/r04f; RAW=/tmp/prof_r04f
mkdir -p $OUT $RAW
python bench.py --steps 20 --warmup 5 2> $OUT/bench.err | tail -1 > $OUT/bench.json
python -m pytest tests/test_fulldepth_gpu.py tests/test_decision_parity_gpu.py tests/test_w8a8_decisions_gpu.py -m gpu -q -s 2>&1 | grep -v "^$" > $OUT/fulldepth_parity_log.txt
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-search-leg --no-small-batch --no-config5-line --no-stream-leg"
rocprofv3 --kernel-trace --stats -d $RAW/stats -o k -- $B --steps 3 --warmup 1 > /dev/null 2>&1
for c in "FETCH_SIZE:f" "WRITE_SIZE:w" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES:m"; do
  n=${c##*:}; ctr=${c%%:*}
  rocprofv3 --pmc $ctr --kernel-trace -d $RAW/pmc_$n -o pmc -- $B --steps 1 --warmup 0 > /dev/null 2>&1 || echo "pass $n failed"
done
rocprofv3 --kernel-trace --stats -d $RAW/vqa -o k -- python $R/tools/vqa_bench.py --batches 1 --steps 48 > /dev/null 2>&1
cd $R
python tools/rocpd_summary.py $RAW/stats/k_results.db > $OUT/kernel_stats.csv
python tools/rocpd_summary.py $RAW/vqa/k_results.db > $OUT/vqa_kernel_stats.csv
python tools/pmc_summary.py $RAW/pmc_f/pmc_results.db $RAW/pmc_w/pmc_results.db $RAW/pmc_m/pmc_results.db > $OUT/pmc.json
python tools/gemm_bench.py --iters 40 > $OUT/gemm_bench.txt 2>/dev/null
python tools/vqa_bench.py --out $OUT/vqa_bench.json > /dev/null 2>&1 || true
python tools/cue_bench.py 2>/dev/null | tail -1 > $OUT/cue_bench.json || true
cp gpurun_out/w8a8_decisions*.json gpurun_out/decision_parity.json $OUT/ 2>/dev/null
ls -la $OUT
