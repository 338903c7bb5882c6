#!/bin/bash
# A/B of two builds of libvstar_hip.so on the LLaMA / ViT GEMM shapes (same box, interleaved runs):
#   tools/ab_gemm.sh /path/libA.so /path/libB.so [rounds]
A=$1; B=$2; R=${3:-3}
for r in $(seq 1 $R); do
  for L in $A $B; do
    echo "== $(basename $L) round $r"
    VSTAR_LIB=$L python tools/gemm_bench.py --iters 40 2>/dev/null | grep -E "llama|clip fc1|owl qkv|owl fc1|square"
  done
done
