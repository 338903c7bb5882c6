#!/bin/bash
# Round-6 evidence: tools/collect_r06.sh [stats|full|fp8]  ->  gpurun_out/r06f/  (summaries only; raw rocprofv3 databases stay in /tmp)
# Every summary is stamped with the hash compiled into the library the profiled command loaded (vstar_amd/provenance.py::checked_hash).
R=$(pwd); OUT=$R/gpurun_out/r06f; RAW=/tmp/prof_r06f
mkdir -p $OUT $RAW
B="python $R/bench.py --no-cpu-baseline --no-search-leg --no-small-batch --no-config5-line --no-stream-leg --no-power-sample"
cd /tmp && export TMPDIR=/tmp
if [ "$1" = "fp8" ]; then      # config 5 (W8A8, 64-crop batches): kernel statistics + the bench line
  rocprofv3 --kernel-trace --stats -d $RAW/stats8 -o k -- $B --fp8 --batch 64 --steps 3 --warmup 1 > /dev/null 2>&1
  cd $R
  python tools/rocpd_summary.py $(ls $RAW/stats8/*/k_results.db $RAW/stats8/k_results.db 2>/dev/null | head -1) > $OUT/kernel_stats_fp8.csv
  for c in "FETCH_SIZE:f" "WRITE_SIZE:w" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES:m"; do
    n=${c##*:}; ctr=${c%%:*}
    (cd /tmp && rocprofv3 --pmc $ctr --kernel-trace -d $RAW/pmc8_$n -o pmc -- $B --fp8 --batch 64 --steps 1 --warmup 0 > /dev/null 2>&1) || echo "fp8 pass $n failed"
  done
  f8() { ls $RAW/pmc8_$1/*/pmc_results.db $RAW/pmc8_$1/pmc_results.db 2>/dev/null | head -1; }
  python tools/pmc_summary.py $(f8 f) $(f8 w) $(f8 m) > $OUT/pmc_fp8.json
  $B --fp8 --batch 64 --steps 10 --warmup 3 2> $OUT/bench_fp8.err | tail -1 > $OUT/bench_fp8.json
  head -16 $OUT/kernel_stats_fp8.csv | cut -c1-180; python -c "
import json; d=json.load(open('$OUT/bench_fp8.json')); print('fp8 bench', d['value'], d['ms_per_step'], d['roofline'])"
  exit 0
fi
rocprofv3 --kernel-trace --stats -d $RAW/stats -o k -- $B --steps 3 --warmup 1 > /dev/null 2>&1
if [ "$1" != "stats" ]; then
for c in "FETCH_SIZE:f" "WRITE_SIZE:w" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES:m"; do
  n=${c##*:}; ctr=${c%%:*}
  rocprofv3 --pmc $ctr --kernel-trace -d $RAW/pmc_$n -o pmc -- $B --steps 1 --warmup 0 > /dev/null 2>&1 || echo "pass $n failed"
done
fi
cd $R
python tools/rocpd_summary.py $(ls $RAW/stats/*/k_results.db $RAW/stats/k_results.db 2>/dev/null | head -1) > $OUT/kernel_stats.csv
if [ "$1" != "stats" ]; then
  f() { ls $RAW/pmc_$1/*/pmc_results.db $RAW/pmc_$1/pmc_results.db 2>/dev/null | head -1; }
  python tools/pmc_summary.py $(f f) $(f w) $(f m) > $OUT/pmc.json
  cp $OUT/pmc.json profiles/r06_pmc_final.json      # (on the box: so that the bench line below quotes THIS build's traffic figure)
  python bench.py --steps 20 --warmup 5 2> $OUT/bench.err | tail -1 > $OUT/bench.json
  python -c "
import json; d=json.load(open('$OUT/bench.json')); r=d['roofline']
print('bench', d['value'], d['ms_per_step'], r['frac'], r.get('under_load'), 'traffic', r['traffic'], r['kernel_source_hash'])"
fi
head -30 $OUT/kernel_stats.csv | cut -c1-180
