#!/bin/bash
# round 3 lab: what changes with one LDS tile per wave (more waves admitted)? PMC passes of the lab sweep, default against POLAR_LP_TILES=1
R=$PWD
cd /tmp && export TMPDIR=/tmp
for var in default tiles1; do
  i=0
  for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
             "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" \
             "TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    if [ $var = tiles1 ]; then export POLAR_LP_TILES=1; else unset POLAR_LP_TILES; fi
    LAB_CASES=5x5x4:fixed LAB_STEPS=2 LAB_KERNELS="lp=" timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/rr_${var}_pmc$i -- python $R/tools/sweep_ab.py > $R/gpurun_out/rr_${var}_pmc$i.log 2>&1
    echo "$var pass $i rc=$?"
  done
done
cd $R
for var in default tiles1; do echo "== $var"; python tools/show_pmc.py rr_${var} k_field_lp; done
