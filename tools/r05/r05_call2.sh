#!/bin/bash
# round 5, GPU call 2: the direct (in-register) gemm256 epilogue — bit-identity tests, per-shape A/B, step A/B
R=$(pwd); OUT=$R/gpurun_out/r05; mkdir -p $OUT
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "gemm" 2>&1 | tail -5 > $OUT/c2_ops_tests.txt
timeout 600 python -m pytest tests/test_engine_gpu.py -m gpu -x -q -k "fused_rope or golden" 2>&1 | tail -5 >> $OUT/c2_ops_tests.txt
cat $OUT/c2_ops_tests.txt
python tools/gemm_bench.py --iters 30 > $OUT/c2_gemm_bench_direct.txt 2>/dev/null
VSTAR_GEMM_DIRECT=0 python tools/gemm_bench.py --iters 30 > $OUT/c2_gemm_bench_lds.txt 2>/dev/null
paste $OUT/c2_gemm_bench_direct.txt $OUT/c2_gemm_bench_lds.txt | awk '{print $0}' | cut -c1-200
B="python bench.py --no-cpu-baseline --no-search-leg --no-small-batch --no-config5-line --no-stream-leg --steps 10 --warmup 3"
$B 2>/dev/null | tail -1 > $OUT/c2_bench_direct.json
VSTAR_GEMM_DIRECT=0 $B 2>/dev/null | tail -1 > $OUT/c2_bench_lds.json
python - <<'PY'
import json
for n in ("direct","lds"):
    try:
        d=json.load(open(f"gpurun_out/r05/c2_bench_{n}.json")); print(n, d["value"], d["ms_per_step"], d["roofline"]["frac"])
    except Exception as e: print(n, "failed", e)
PY
