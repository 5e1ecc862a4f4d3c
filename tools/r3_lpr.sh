#!/bin/bash
# round 3 lab: k_field_lpr (R launch rows per wave, next row's start hidden behind the current row's trips) against k_field_lp
LAB_CASES="${LAB_CASES:-3x3x3:fixed,5x5x4:prec}" LAB_STEPS=4 LAB_KERNELS="${LAB_KERNELS:-lp=,r2=POLAR_LP_ROWS=2,r3=POLAR_LP_ROWS=3,r4=POLAR_LP_ROWS=4,r6=POLAR_LP_ROWS=6,r8=POLAR_LP_ROWS=8}" timeout -k 10 600 python tools/sweep_ab.py > gpurun_out/r3_lpr.log 2>&1
grep -v "amdgpu.ids" gpurun_out/r3_lpr.log | tail -16
