#!/bin/bash
# the whole GPU suite, then config 0 alone (bare)
mkdir -p gpurun_out
tag=${1:-r4u}
timeout -k 10 1000 python -m pytest tests -q -m gpu > gpurun_out/${tag}_tests.log 2>&1
echo "tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|Aborted" gpurun_out/${tag}_tests.log | tail -10
timeout -k 10 300 python tools/r4_c0.py > gpurun_out/${tag}_c0.txt 2>&1
echo "c0 rc=$?"; grep config0 gpurun_out/${tag}_c0.txt
timeout -k 10 600 python tools/r4_x10k.py > gpurun_out/${tag}_x10k.txt 2>&1; echo "x10k rc=$?"; tail -1 gpurun_out/${tag}_x10k.txt | cut -c1-600
