#!/bin/bash
# round 5, GPU call 4: tile timeline (direct vs LDS epilogue), fixed ln_fold test, the x32 parity test + the six fixtures
R=$(pwd); OUT=$R/gpurun_out/r05; mkdir -p $OUT
VSTAR_LIB=$R/vstar_amd/csrc/build/ab/lib_tl.so python tools/gemm_timeline.py > $OUT/gemm_timeline_direct.txt 2>&1
VSTAR_GEMM_DIRECT=0 VSTAR_LIB=$R/vstar_amd/csrc/build/ab/lib_tl.so python tools/gemm_timeline.py > $OUT/gemm_timeline_lds.txt 2>&1
cat $OUT/gemm_timeline_direct.txt; echo ---- LDS; grep -v "^$" $OUT/gemm_timeline_lds.txt | grep -v ticks
timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "ln_fold" 2>&1 | tail -3
timeout 1500 python -m pytest tests/test_fulldepth_gpu.py -m gpu -q -s 2>&1 | grep -v "^$" > $OUT/fulldepth_parity_log.txt; tail -25 $OUT/fulldepth_parity_log.txt | cut -c1-250
