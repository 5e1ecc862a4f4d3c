#!/bin/bash
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "deterministic or polar_sor or product_library" > gpurun_out/r3_det2_tests.log 2>&1; tail -3 gpurun_out/r3_det2_tests.log
LAB_LIB=0 LAB_CASES="3x3x3:fixed,5x5x4:prec" LAB_STEPS=5 LAB_KERNELS="default=,det=kw:deterministic:1,sor1.15=kw:polar_sor:1.15,det+sor1.15=kw:deterministic:1;kw:polar_sor:1.15" timeout -k 10 700 python tools/sweep_ab.py > gpurun_out/r3_det_sor.log 2>&1
grep -v "colour phases\|amdgpu.ids" gpurun_out/r3_det_sor.log | tail -12
