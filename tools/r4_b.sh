#!/bin/bash
# round 4: the new multi-GPU schedule on the one-GPU box -- tests over the mock transport, set_positions parity, then the emulations
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "in_library or set_positions or sweep_in_parts" > gpurun_out/r4b_tests.log 2>&1
echo "tests rc=$?" | tee -a gpurun_out/r4b_tests.log
grep -E "^(FAILED|ERROR)|passed|failed|AssertionError" gpurun_out/r4b_tests.log | tail -30
