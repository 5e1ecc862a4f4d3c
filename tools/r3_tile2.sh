#!/bin/bash
# round 3 lab: tile sweep variants (waves per workgroup, wide cells) against the row sweep
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x "tests/test_gpu_parity.py::test_alternative_sweep_kernels_agree[POLAR_SWEEP_KERNEL=4]" "tests/test_gpu_parity.py::test_alternative_sweep_kernels_agree[POLAR_SWEEP_KERNEL=4;POLAR_DETERMINISTIC=1]" > gpurun_out/r3_tile2_tests.log 2>&1
tail -3 gpurun_out/r3_tile2_tests.log
LAB_CASES="${LAB_CASES:-3x3x3:fixed,5x5x4:prec}" LAB_STEPS=3 LAB_KERNELS="${LAB_KERNELS:-lp=POLAR_SWEEP_KERNEL=2,tile4=POLAR_SWEEP_KERNEL=4,tile8=POLAR_SWEEP_KERNEL=4;POLAR_TILE_WAVES=8,wide4=POLAR_SWEEP_KERNEL=4;POLAR_TILE_WIDE=1,wide8=POLAR_SWEEP_KERNEL=4;POLAR_TILE_WIDE=1;POLAR_TILE_WAVES=8}" POLAR_DEBUG=1 timeout -k 10 500 python tools/sweep_ab.py > gpurun_out/r3_tile2_ab.log 2>&1
cat gpurun_out/r3_tile2_ab.log | grep -v "colour phases" | tail -24
