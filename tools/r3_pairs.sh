#!/bin/bash
# round 3 lab: paired rows (k_field_lp2: two rows of a colour phase that share a cell swept by one wave over the union of their neighbours)
LAB_CASES="${LAB_CASES:-3x3x3:fixed,5x5x4:prec}" LAB_STEPS=4 POLAR_DEBUG=${POLAR_DEBUG:-} LAB_KERNELS="${LAB_KERNELS:-lp=,pairs=POLAR_LP_PAIRS=1}" timeout -k 10 600 python tools/sweep_ab.py > gpurun_out/r3_pairs.log 2>&1
grep -v "amdgpu.ids" gpurun_out/r3_pairs.log | tail -12
