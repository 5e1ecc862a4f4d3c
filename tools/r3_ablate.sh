#!/bin/bash
# round 3 lab: what the row sweep waits for -- k_field_lp with parts switched off (lab library, POLAR_ABLATE bits:
# 1 rows without trips, 4 every gather hits record 0, 8 no gathers at all, 16 no pair arithmetic); timing only, wrong numbers
LAB_CASES="${LAB_CASES:-5x5x4:fixed}" LAB_STEPS=4 LAB_KERNELS="full=,rows_only=POLAR_ABLATE=1,gathers_hit_one_record=POLAR_ABLATE=4,no_gathers=POLAR_ABLATE=8,no_arithmetic=POLAR_ABLATE=16,no_gathers_no_arithmetic=POLAR_ABLATE=24" timeout -k 10 600 python tools/sweep_ab.py > gpurun_out/r3_ablate.log 2>&1
grep -v "amdgpu.ids" gpurun_out/r3_ablate.log | tail -8
