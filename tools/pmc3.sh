#!/bin/bash
# PMC passes for the sweep kernel (memory-path counters); one group per run, CSV under gpurun_out/<tag>_pmcN
tag=${1:-pmc3}; shift
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $R/gpurun_out/${tag}_avail.txt 2>&1
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD" \
           "FETCH_SIZE TCC_HIT_sum TCC_MISS_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
           "TA_TA_BUSY_sum TA_BUSY_avr" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
           "TCC_REQ_sum TCC_READ_sum TCC_BUSY_avr" \
           "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/${tag}_pmc$i -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $R/gpurun_out/${tag}_pmc$i.log 2>&1
  echo "pass $i ($grp) rc=$?"
done
