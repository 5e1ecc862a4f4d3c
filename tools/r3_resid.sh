#!/bin/bash
# round 3 lab: residency knobs of k_field_lp on the current kernel (one LDS tile per wave, workgroup sizes)
LAB_CASES="${LAB_CASES:-5x5x4:fixed}" LAB_STEPS=4 LAB_KERNELS="lp=,tiles1=POLAR_LP_TILES=1,tiles1_b128=POLAR_LP_TILES=1;POLAR_QUAD_BLOCK=128,b64=POLAR_QUAD_BLOCK=64,b128=POLAR_QUAD_BLOCK=128,b512=POLAR_QUAD_BLOCK=512" timeout -k 10 600 python tools/sweep_ab.py > gpurun_out/r3_resid.log 2>&1
grep -v "amdgpu.ids" gpurun_out/r3_resid.log | tail -8
