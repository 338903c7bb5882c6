#!/bin/bash
# FETCH_SIZE / WRITE_SIZE / LDS instruction count per launch of hipBLASLt, gemm256, gemm4w on the LLaMA shapes (one counter set per pass)
R=$(pwd); RAW=/tmp/prof_r06t; mkdir -p $RAW $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
ARGS=""
for sh in ${SHAPES:-qkv gate_up_silu down}; do
  for c in "FETCH_SIZE:f" "WRITE_SIZE:w" "SQ_INSTS_LDS:l"; do
    n=${c##*:}; ctr=${c%%:*}
    timeout 300 rocprofv3 --pmc $ctr --kernel-trace -d $RAW/${sh}_$n -o pmc -- python $R/tools/gemm_traffic.py run $sh 3 > /dev/null 2>&1 || echo "pass $sh $n failed"
    ARGS="$ARGS $sh=$(ls $RAW/${sh}_$n/*/pmc_results.db $RAW/${sh}_$n/pmc_results.db 2>/dev/null | head -1)"
  done
done
cd $R
python tools/gemm_traffic.py summary $ARGS
