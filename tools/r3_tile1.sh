#!/bin/bash
# round 3 lab: the tile sweep -- parity of its variants, then A/B against the row sweep
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x "tests/test_gpu_parity.py::test_alternative_sweep_kernels_agree[POLAR_SWEEP_KERNEL=4]" "tests/test_gpu_parity.py::test_alternative_sweep_kernels_agree[POLAR_SWEEP_KERNEL=4;POLAR_DETERMINISTIC=1]" > gpurun_out/r3_tile1_tests.log 2>&1
tail -15 gpurun_out/r3_tile1_tests.log
LAB_CASES="${LAB_CASES:-3x3x3:fixed,5x5x4:prec}" LAB_KERNELS="${LAB_KERNELS:-lp=POLAR_SWEEP_KERNEL=2,tile=POLAR_SWEEP_KERNEL=4,tiledet=POLAR_SWEEP_KERNEL=4;POLAR_DETERMINISTIC=1}" POLAR_DEBUG=1 timeout -k 10 500 python tools/sweep_ab.py > gpurun_out/r3_tile1_ab.log 2>&1
cat gpurun_out/r3_tile1_ab.log | tail -20
