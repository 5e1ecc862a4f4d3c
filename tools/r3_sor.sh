#!/bin/bash
# round 3 lab: over-relaxed colour-phase Gauss-Seidel (POLAR_SOR) and workgroup sizes of the row sweep
LAB_CASES="${LAB_CASES:-3x3x3:prec,5x5x4:prec}" LAB_STEPS=3 LAB_KERNELS="${LAB_KERNELS:-w1.00=POLAR_SOR=1.0,w1.05=POLAR_SOR=1.05,w1.10=POLAR_SOR=1.10,w1.15=POLAR_SOR=1.15,w1.20=POLAR_SOR=1.20,w1.25=POLAR_SOR=1.25,w1.30=POLAR_SOR=1.30,b64=POLAR_QUAD_BLOCK=64,b128=POLAR_QUAD_BLOCK=128,b512=POLAR_QUAD_BLOCK=512}" timeout -k 10 700 python tools/sweep_ab.py > gpurun_out/r3_sor_ab.log 2>&1
cat gpurun_out/r3_sor_ab.log | grep -v "colour phases\|amdgpu.ids" | tail -30
