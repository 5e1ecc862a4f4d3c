#!/bin/bash
mkdir -p gpurun_out; tag=${1:-r4sh}
R=$PWD; cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_prof -- python $R/tools/r4_shard_prof.py > $R/gpurun_out/${tag}.log 2>&1
echo "rc=$?"; cd $R; tail -3 gpurun_out/${tag}.log
python3 - <<PY
import csv,glob
f=glob.glob('gpurun_out/${tag}_prof/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:28]:
    print(r['Name'][:60].ljust(60), r['Calls'].rjust(5), ('%.1f'%(float(r['AverageNs'])/1e3)).rjust(9),'us', ('%.3f'%(int(r['TotalDurationNs'])/6/1e6)).rjust(7),'ms/step')
PY
