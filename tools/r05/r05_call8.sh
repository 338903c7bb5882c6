#!/bin/bash
R=$(pwd); OUT=$R/gpurun_out/r05; mkdir -p $OUT
AB=$R/vstar_amd/csrc/build/ab
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "gemm" 2>&1 | tail -4
for i in 1 2; do
  python tools/gemm_bench.py --iters 30 > $OUT/c8_gemm_new_$i.txt 2>/dev/null
  VSTAR_LIB=$AB/lib_ring8.so python tools/gemm_bench.py --iters 30 > $OUT/c8_gemm_ring8_$i.txt 2>/dev/null
  VSTAR_LIB=$AB/lib_r04gemm.so python tools/gemm_bench.py --iters 30 > $OUT/c8_gemm_r04_$i.txt 2>/dev/null
done
python - <<'PY'
import re
def rd(p):
    out={}
    for l in open(p):
        m=re.match(r"(.+?)\s+(\d+)\s+(\d+)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s*$", l)
        if m: out[m.group(1).strip()]=float(m.group(6))
    return out
g=lambda t:[rd(f"gpurun_out/r05/c8_gemm_{t}_{i}.txt") for i in (1,2)]
a,b,c=g("new"),g("ring8"),g("r04")
print(f"{'shape':22s} {'ring10':>8s} {'ring8':>8s} {'r04':>8s}  ring10/r04")
for k in a[0]:
    n=max(x[k] for x in a); o=max(x.get(k,0) for x in b); r=max(x.get(k,0) for x in c)
    print(f"{k:22s} {n:8.1f} {o:8.1f} {r:8.1f}  x{n/max(r,1e-9):.3f}")
PY
