#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "polar_accel" > gpurun_out/r4d_tests.log 2>&1
echo "tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|^E  " gpurun_out/r4d_tests.log | tail -20
timeout -k 10 600 python tools/r4_accel.py 2 > gpurun_out/r4d_accel.txt 2>&1
echo "accel rc=$?"; grep -v amdgpu.ids gpurun_out/r4d_accel.txt
