#!/bin/bash
# round 5: POLAR_POISON=1 (every new device buffer filled with 0x7F bytes, fill synchronised): the edge-case file first, then --
# only if that was clean -- the whole GPU suite ONCE.  --capture=sys keeps a message of the HSA runtime on fd 2 in the log.
tag=${1:-r5d}
mkdir -p gpurun_out
POLAR_POISON=1 timeout -k 10 600 python -m pytest tests/test_gpu_edges.py -q --capture=sys --tb=short > gpurun_out/${tag}_edges.log 2>&1
rc=$?
echo "edges rc=$rc"; grep -E "^(FAILED|ERROR)|passed|failed|Aborted|fault" gpurun_out/${tag}_edges.log | head -20
if grep -q "Memory access fault" gpurun_out/${tag}_edges.log; then exit 9; fi
if [ $rc -ne 0 ]; then exit $rc; fi
POLAR_POISON=1 timeout -k 10 1000 python -m pytest tests -q -m gpu --capture=sys --tb=short > gpurun_out/${tag}_tests.log 2>&1
rc=$?
echo "suite rc=$rc"; grep -E "^(FAILED|ERROR)|passed|failed|Aborted|fault" gpurun_out/${tag}_tests.log | head -20
if grep -q "Memory access fault" gpurun_out/${tag}_tests.log; then exit 9; fi
exit $rc
