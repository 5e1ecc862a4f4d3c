#!/bin/bash
# round 3 lab: device colouring (Jones-Plassmann, saturation-first, + folding) against the host-side DSATUR: phases, sweeps
LAB_CASES="${LAB_CASES:-3x3x3:prec,5x5x4:prec}" LAB_STEPS=3 POLAR_DEBUG=1 LAB_KERNELS="${LAB_KERNELS:-cells=,jp=POLAR_COLOR_JP=1,host=POLAR_HOST_COLORS=1}" timeout -k 10 600 python tools/sweep_ab.py > gpurun_out/r3_colors.log 2>&1
grep -v "amdgpu.ids" gpurun_out/r3_colors.log | tail -12
