#!/bin/bash
# round 5: the tests that FAILED BY ASSERTION (no fault) under POLAR_POISON=1, alone, with their assertion messages
tag=${1:-r5c}
mkdir -p gpurun_out
POLAR_POISON=1 timeout -k 10 600 python -m pytest tests/test_gpu_edges.py -q --capture=sys --tb=short \
  -k "small_random_system or gs-13 or jacobi5-14 or ranked-12" > gpurun_out/${tag}_tests.log 2>&1
rc=$?
echo "tests rc=$rc"; grep -E "^(FAILED|ERROR)|passed|failed|Aborted|fault|assert|Error" gpurun_out/${tag}_tests.log | head -40
if grep -q "Memory access fault" gpurun_out/${tag}_tests.log; then exit 9; fi
exit 0
