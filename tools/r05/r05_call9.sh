#!/bin/bash
R=$(pwd); OUT=$R/gpurun_out/r05; mkdir -p $OUT
timeout 600 python -m pytest tests/test_score_graph_gpu.py -m gpu -x -q 2>&1 | tail -15
for b in 1 2 4 8; do
  for g in 1 0; do
    VSTAR_SCORE_GRAPH=$g python bench.py --batch $b --steps 30 --warmup 5 --no-cpu-baseline --no-search-leg --no-small-batch --no-config5-line --no-stream-leg 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('B=$b graph=$g', d['ms_per_step'], 'ms', d['value'], 'crops/s')"
  done
done | tee $OUT/c9_small_batch_graph_ab.txt
