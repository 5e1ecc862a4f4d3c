#!/bin/bash
mkdir -p gpurun_out; tag=${1:-r4tr}
R=$PWD; cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_prof -- python $R/tools/r4_dist_trace.py > $R/gpurun_out/${tag}.log 2>&1
echo "rc=$?"; cd $R; tail -1 gpurun_out/${tag}.log
python3 - <<PY
import csv,glob
f=glob.glob('gpurun_out/${tag}_prof/*/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# the last step: take the last 400 kernels, print a window of 40 in the middle
w=rows[-330:-290]
t0=int(w[0]['Start_Timestamp'])
for r in w:
    print(r['Kernel_Name'][:44].ljust(44), 'q', r.get('Queue_Id','?'), 'start %8.1f us  dur %6.1f us' % ((int(r['Start_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3))
PY
