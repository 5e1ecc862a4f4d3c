#!/bin/bash
# round 4: the fuzz system that aborted under the region pipeline, once more with and without it (stderr kept), then the suite
mkdir -p gpurun_out
POLAR_PIPELINE=0 timeout -k 10 200 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -k "oracle and (30 or 31)" > gpurun_out/r4g_fuzz_p0.log 2>&1
echo "pipeline 0 rc=$?"; tail -3 gpurun_out/r4g_fuzz_p0.log
POLAR_PIPELINE=1 POLAR_DEBUG=1 timeout -k 10 200 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -x -k "oracle and 31" > gpurun_out/r4g_fuzz_p1.log 2>&1
rc=$?; echo "pipeline 1 rc=$rc"; grep -v "^  File" gpurun_out/r4g_fuzz_p1.log | tail -25
if [ $rc -ne 0 ]; then echo "stopping after the failed GPU step"; exit 0; fi
timeout -k 10 1000 python -m pytest tests -q -m gpu > gpurun_out/r4g_tests.log 2>&1
echo "tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|Aborted" gpurun_out/r4g_tests.log | tail -20
