#!/bin/bash
tag=${1:-pmc2}; shift
R=$PWD
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_IFETCH SQ_INSTS_FLAT" \
           "GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/${tag}_pmc$i -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $R/gpurun_out/${tag}_pmc$i.log 2>&1
  echo "pass $i rc=$?"
done
