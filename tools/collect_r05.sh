#!/bin/bash
# Round-5 evidence: tools/collect_r05.sh [quick]  ->  gpurun_out/r05f/  (summaries only; raw rocprofv3 databases stay in /tmp)
# Every summary is stamped with the hash of the kernel sources it was measured on (vstar_amd/provenance.py).
R=$(pwd); OUT=$R/gpurun_out/r05f; RAW=/tmp/prof_r05f
mkdir -p $OUT $RAW
B="python $R/bench.py --no-cpu-baseline --no-search-leg --no-small-batch --no-config5-line --no-stream-leg --no-power-sample"
cd /tmp && export TMPDIR=/tmp
# ---- bf16 headline: kernel trace + the three PMC passes (separate runs, per the guide) ----
rocprofv3 --kernel-trace --stats -d $RAW/stats -o k -- $B --steps 3 --warmup 1 > /dev/null 2>&1
for c in "FETCH_SIZE:f" "WRITE_SIZE:w" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES:m"; do
  n=${c##*:}; ctr=${c%%:*}
  rocprofv3 --pmc $ctr --kernel-trace -d $RAW/pmc_$n -o pmc -- $B --steps 1 --warmup 0 > /dev/null 2>&1 || echo "pass $n failed"
done
# ---- W8A8 (config 5 precision), 64-crop batches: the same four passes ----
F="$B --fp8 --batch 64"
[ "$1" != "bf16" ] && rocprofv3 --kernel-trace --stats -d $RAW/stats8 -o k -- $F --steps 3 --warmup 1 > /dev/null 2>&1
if [ "$1" != "quick" ] && [ "$1" != "bf16" ]; then
for c in "FETCH_SIZE:f" "WRITE_SIZE:w" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES:m"; do
  n=${c##*:}; ctr=${c%%:*}
  rocprofv3 --pmc $ctr --kernel-trace -d $RAW/pmc8_$n -o pmc -- $F --steps 1 --warmup 0 > /dev/null 2>&1 || echo "fp8 pass $n failed"
done
fi
cd $R
python tools/rocpd_summary.py $RAW/stats/k_results.db > $OUT/kernel_stats.csv
[ "$1" != "bf16" ] && python tools/rocpd_summary.py $RAW/stats8/k_results.db > $OUT/kernel_stats_fp8.csv
python tools/pmc_summary.py $RAW/pmc_f/pmc_results.db $RAW/pmc_w/pmc_results.db $RAW/pmc_m/pmc_results.db > $OUT/pmc.json
cp $OUT/pmc.json profiles/r05_pmc_final.json      # (on the box: so that the bench line below quotes THIS build's traffic figure)
[ "$1" != "quick" ] && [ "$1" != "bf16" ] && python tools/pmc_summary.py $RAW/pmc8_f/pmc_results.db $RAW/pmc8_w/pmc_results.db $RAW/pmc8_m/pmc_results.db > $OUT/pmc_fp8.json
[ "$1" != "bf16" ] && head -6 $OUT/kernel_stats_fp8.csv | cut -c1-200
if [ "$1" = "bf16" ]; then      # (round 5, last kernel edit touched decode.hip only: re-stamp the bf16 evidence, bench line, decode benches)
  python bench.py --steps 20 --warmup 5 2> $OUT/bench.err | tail -1 > $OUT/bench.json
  python tools/vqa_bench.py --out $OUT/vqa_bench.json > /dev/null 2>&1 || true
  python tools/cue_bench.py 2>/dev/null | tail -1 > $OUT/cue_bench.json || true
  python -c "
import json; d=json.load(open('$OUT/bench.json')); r=d['roofline']
print('bench', d['value'], d['ms_per_step'], r['frac'], r['under_load'], 'traffic', r['traffic'], r['kernel_source_hash'])
print('config5', d['config5'].get('crops_per_s'), d['config5']['roofline']['fp8_linears'])"
fi
if [ "$1" != "quick" ] && [ "$1" != "bf16" ]; then
  # ---- the driver's line, per-shape GEMM table, parity logs, decode benches ----
  python bench.py --steps 20 --warmup 5 2> $OUT/bench.err | tail -1 > $OUT/bench.json
  python tools/gemm_bench.py --iters 40 > $OUT/gemm_bench.txt 2>/dev/null
  python -m pytest tests/test_fulldepth_gpu.py tests/test_decision_parity_gpu.py tests/test_w8a8_decisions_gpu.py -m gpu -q -s 2>&1 | grep -v "^$" > $OUT/fulldepth_parity_log.txt
  for g in 336 224; do NOISE_STUDY_GEOMETRY=$g python tools/noise_study.py 2>/dev/null | grep NOISE_STUDY | sed "s/^NOISE_STUDY //" > $OUT/noise_study_x32_$g.json; done
  python tools/vqa_bench.py --out $OUT/vqa_bench.json > /dev/null 2>&1 || true
  python tools/cue_bench.py 2>/dev/null | tail -1 > $OUT/cue_bench.json || true
  cp gpurun_out/w8a8_decisions*.json gpurun_out/decision_parity.json $OUT/ 2>/dev/null
  python -c "
import json; d=json.load(open('$OUT/bench.json')); r=d['roofline']
print('bench', d['value'], d['ms_per_step'], r['frac'], r.get('under_load'), 'traffic', r['traffic'])
print('config5', d['config5'].get('crops_per_s'), d['config5'].get('roofline'))"
  tail -4 $OUT/fulldepth_parity_log.txt
fi
ls -la $OUT
