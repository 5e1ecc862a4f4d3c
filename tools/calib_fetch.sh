#!/bin/bash
# FETCH_SIZE calibration run (tools/calib_fetch.hip): separate --pmc passes, per-kernel counter means printed
R=$PWD
cd /tmp && export TMPDIR=/tmp
for grp in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  d=$R/gpurun_out/calib_$(echo $grp | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $d -- $R/tools/calib_fetch > $d.log 2>&1
  echo "pass ($grp) rc=$?"
done
cd $R
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/calib_*/**/*counter_collection.csv', recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[(r['Kernel_Name'][:24], r['Counter_Name'], r.get('Dispatch_Id'))].append(float(r['Counter_Value']))
    per = collections.defaultdict(list)
    for (k, c, d), v in acc.items():
        per[(k, c)].append(sum(v))
    for (k, c), v in sorted(per.items()):
        print(f"{k:26s} {c:24s} per dispatch: {[round(x, 1) for x in v]}")
PY
