#!/bin/bash
# round 4: config 0 with the block-inverse exact-order sweep: parity (the golden tests), timing bare and under rocprofv3
mkdir -p gpurun_out
tag=${1:-r4m}
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu > gpurun_out/${tag}_tests.log 2>&1
echo "tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|Aborted" gpurun_out/${tag}_tests.log | tail -10
timeout -k 10 300 python tools/r4_c0.py > gpurun_out/${tag}_c0.txt 2>&1 && timeout -k 10 300 python tools/r4_c0.py >> gpurun_out/${tag}_c0.txt 2>&1
echo "c0 rc=$?"; cat gpurun_out/${tag}_c0.txt | tail -4
R=$PWD; cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_c0prof -- python $R/tools/r4_c0.py > $R/gpurun_out/${tag}_c0prof.log 2>&1
echo "prof rc=$?"; cd $R
cut -d, -f1-4 gpurun_out/${tag}_c0prof/*/*kernel_stats.csv | cut -c1-110 | head -12
