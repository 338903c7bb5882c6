#!/bin/bash
# SQ / MFMA counters of the attention kernels alone (tools/attn_kernel_bench.py under rocprofv3 --pmc, one counter set per pass).
#   [VSTAR_LIB=...] tools/attn_pmc.sh <tag>  ->  gpurun_out/<tag>_attn_counters.json
TAG=${1:-attn}
R=$(pwd); RAW=/tmp/attnpmc_$TAG; mkdir -p $RAW $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python $R/tools/attn_kernel_bench.py owl llama"
for c in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES:m" \
         "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA:s1" \
         "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU:s2" \
         "SQ_WAVES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES:s3"; do
  n=${c##*:}; ctr=${c%%:*}
  rocprofv3 --pmc $ctr --kernel-trace -d $RAW/pmc_$n -o pmc -- $B > /dev/null 2>&1 || echo "pass $n failed"
done
cd $R
python tools/pmc_dump.py $RAW/pmc_m/pmc_results.db $RAW/pmc_s1/pmc_results.db $RAW/pmc_s2/pmc_results.db $RAW/pmc_s3/pmc_results.db > gpurun_out/${TAG}_attn_counters.json
