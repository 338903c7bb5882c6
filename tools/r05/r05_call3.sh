#!/bin/bash
# round 5, GPU call 3: new op tests, the 32-crop noise study sweep, hipBLASLt kernel names
R=$(pwd); OUT=$R/gpurun_out/r05; mkdir -p $OUT
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "ln_fold or direct_epilogue" 2>&1 | tail -8 > $OUT/c3_tests.txt; cat $OUT/c3_tests.txt
timeout 1500 python tools/noise_study.py --sweep --out $OUT/noise_study.json 2>&1 | tail -12
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /tmp/hbl -o k -- python $R/tools/hipblaslt_kernel_names.py > /dev/null 2>&1
cd $R; python tools/rocpd_summary.py /tmp/hbl/k_results.db > $OUT/hipblaslt_kernels.csv 2>&1; cut -c1-300 $OUT/hipblaslt_kernels.csv | head -12
