#!/bin/bash
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3_ball_prof -- python $R/tools/r3_ballistic.py > $R/gpurun_out/r3_ball.log 2>&1
echo rc=$?
cd $R
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r3_ball_prof/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:40]:
    if 'color' in r['Name'] or 'sort_small' in r['Name'] or 'scan' in r['Name']:
        print(r['Name'][:50].ljust(50), r['Calls'].rjust(6), ('%.1f'%(float(r['AverageNs'])/1e3)).rjust(10),'us  total ms', '%.2f'%(float(r['TotalDurationNs'])/1e6))
PY
tail -1 gpurun_out/r3_ball.log | cut -c1-400
