#!/bin/bash
# GPU tests, then a profiled bench; everything lands under gpurun_out/<tag>
tag=${1:-run}
timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/${tag}_tests.log 2>&1
tail -3 gpurun_out/${tag}_tests.log
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_prof -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/${tag}_bench.log 2>&1
grep '"metric"' $R/gpurun_out/${tag}_bench.log | cut -c1-200
