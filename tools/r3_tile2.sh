#!/bin/bash
LAB_CASES="${LAB_CASES:-5x5x4:prec}" LAB_STEPS=3 LAB_KERNELS="${LAB_KERNELS:-tile=POLAR_SWEEP_KERNEL=4,tilepad=POLAR_SWEEP_KERNEL=4;POLAR_TILE_LDS_PAD=70000,tilepad20=POLAR_SWEEP_KERNEL=4;POLAR_TILE_LDS_PAD=12000}" timeout -k 10 500 python tools/sweep_ab.py > gpurun_out/r3_tile2_ab.log 2>&1
cat gpurun_out/r3_tile2_ab.log | grep -v "^\[polar\]" | tail -20
