#!/bin/bash
R=$(pwd); OUT=$R/gpurun_out/r05; mkdir -p $OUT
R04=$R/vstar_amd/csrc/build/ab/lib_r04gemm.so
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "gemm or ln_fold" 2>&1 | tail -3 > $OUT/c7_tests.txt
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_shared_prefix_gpu.py tests/test_grouped_gpu.py -m gpu -q 2>&1 | tail -6 >> $OUT/c7_tests.txt
cat $OUT/c7_tests.txt
for i in 1 2; do
  python tools/gemm_bench.py --iters 30 > $OUT/c7_gemm_new_$i.txt 2>/dev/null
  VSTAR_LIB=$R04 python tools/gemm_bench.py --iters 30 > $OUT/c7_gemm_r04_$i.txt 2>/dev/null
done
python - <<'PY'
import re
def rd(p):
    out={}
    for l in open(p):
        m=re.match(r"(.+?)\s+(\d+)\s+(\d+)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s*$", l)
        if m: out[m.group(1).strip()]=float(m.group(6))
    return out
a=[rd(f"gpurun_out/r05/c7_gemm_new_{i}.txt") for i in (1,2)]; b=[rd(f"gpurun_out/r05/c7_gemm_r04_{i}.txt") for i in (1,2)]
print(f"{'shape':22s} {'new TF':>8s} {'r04 TF':>8s}  ratio")
for k in a[0]:
    n=max(x[k] for x in a); o=max(x.get(k,0) for x in b)
    print(f"{k:22s} {n:8.1f} {o:8.1f}  x{n/max(o,1e-9):.3f}")
PY
VSTAR_LIB=$R/vstar_amd/csrc/build/ab/lib_tl.so python tools/gemm_timeline.py 2>/dev/null | grep -E "^[a-z]|wg 0 " > $OUT/gemm_timeline_v3.txt; cat $OUT/gemm_timeline_v3.txt
B="python bench.py --no-cpu-baseline --no-search-leg --no-small-batch --no-config5-line --no-stream-leg --steps 10 --warmup 3"
$B 2>/dev/null | tail -1 > $OUT/c7_bench_new.json
VSTAR_LIB=$R04 $B 2>/dev/null | tail -1 > $OUT/c7_bench_r04.json
$B --fp8 --batch 64 2>/dev/null | tail -1 > $OUT/c7_bench_fp8_new.json
VSTAR_LIB=$R04 $B --fp8 --batch 64 2>/dev/null | tail -1 > $OUT/c7_bench_fp8_r04.json
python - <<'PY'
import json
for n in ("new","r04","fp8_new","fp8_r04"):
    try:
        d=json.load(open(f"gpurun_out/r05/c7_bench_{n}.json")); print(n, d["value"], d["ms_per_step"], d["roofline"]["frac"])
    except Exception as e: print(n, "failed", e)
PY
