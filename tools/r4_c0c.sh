#!/bin/bash
# round 4: config 0 quick loop: exact-mode goldens, timing bare and under rocprofv3
mkdir -p gpurun_out
tag=${1:-r4n}
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "golden or exact or matrix_free or config0 or knife" > gpurun_out/${tag}_tests.log 2>&1
echo "tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|Aborted" gpurun_out/${tag}_tests.log | tail -10
timeout -k 10 300 python tools/r4_c0.py > gpurun_out/${tag}_c0.txt 2>&1 && timeout -k 10 300 python tools/r4_c0.py >> gpurun_out/${tag}_c0.txt 2>&1
echo "c0 rc=$?"; grep config0 gpurun_out/${tag}_c0.txt
R=$PWD; cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_c0prof -- python $R/tools/r4_c0.py > $R/gpurun_out/${tag}_c0prof.log 2>&1
echo "prof rc=$?"; cd $R
python3 - <<PY
import csv,glob
f=glob.glob('gpurun_out/${tag}_c0prof/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:6]:
    print(r['Name'][:50].ljust(50), r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
PY
