#!/bin/bash
R=$(pwd); OUT=$R/gpurun_out/r05; mkdir -p $OUT
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "ln_fold" 2>&1 | grep -E "^E|assert|Error" | head -20
timeout 600 python -m pytest tests/test_engine_gpu.py -m gpu -x -q -k "batched_vs_oracle" 2>&1 | grep -E "^E|assert|Error|passed|failed" | head -30
for s in 0 2 4 8; do
  echo "== VSTAR_GEMM_XCD_STAGGER=$s"
  VSTAR_GEMM_XCD_STAGGER=$s python tools/gemm_bench.py --iters 30 2>/dev/null | grep -E "llama|clip fc1|owl out|owl fc1|owl qkv"
done
