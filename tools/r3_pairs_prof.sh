#!/bin/bash
R=$PWD
cd /tmp && export TMPDIR=/tmp
export POLAR_LP_PAIRS=1
LAB_CASES=5x5x4:prec LAB_STEPS=3 LAB_KERNELS="pairs=POLAR_LP_PAIRS=1" timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3_pairs_prof -- python $R/tools/sweep_ab.py > $R/gpurun_out/r3_pairs_prof.log 2>&1
echo rc=$?
cd $R
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r3_pairs_prof/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(r['Name'][:48].ljust(48), r['Calls'].rjust(6), ('%.1f'%(float(r['AverageNs'])/1e3)).rjust(10),'us  %', r['Percentage'])
PY
