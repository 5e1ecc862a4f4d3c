#!/bin/bash
# the whole GPU suite, then the full default bench line
mkdir -p gpurun_out
tag=${1:-r4x}
timeout -k 10 1000 python -m pytest tests -q -m gpu > gpurun_out/${tag}_tests.log 2>&1
echo "tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|Aborted" gpurun_out/${tag}_tests.log | tail -10
timeout -k 10 600 python bench.py > gpurun_out/${tag}_bench_full.log 2>&1
echo "bench rc=$?"; tail -1 gpurun_out/${tag}_bench_full.log | cut -c1-200
