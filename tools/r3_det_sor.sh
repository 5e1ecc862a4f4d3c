#!/bin/bash
# round 3 lab: cost of `deterministic yes`, and `polar_sor`, on configs[1] and configs[2] (PRODUCT library: LAB_LIB=0)
LAB_LIB=0 LAB_CASES="${LAB_CASES:-3x3x3:fixed,5x5x4:prec}" LAB_STEPS=5 LAB_KERNELS="${LAB_KERNELS:-default=,det=kw:deterministic:1,sor1.10=kw:polar_sor:1.10,sor1.15=kw:polar_sor:1.15,sor1.20=kw:polar_sor:1.20,det+sor1.15=kw:deterministic:1;kw:polar_sor:1.15}" timeout -k 10 700 python tools/sweep_ab.py > gpurun_out/r3_det_sor.log 2>&1
grep -v "colour phases\|amdgpu.ids" gpurun_out/r3_det_sor.log | tail -20
