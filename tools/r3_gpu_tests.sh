#!/bin/bash
# the whole GPU suite in one process (log under gpurun_out/)
tag=${1:-r3}
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/${tag}_tests.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/${tag}_tests.log | cut -c1-300
