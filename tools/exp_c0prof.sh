#!/bin/bash
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/c0_prof -- python $R/tools/config0_timing.py > $R/gpurun_out/c0_prof.log 2>&1
cd $R
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/c0_prof/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:10]:
    print(r["Name"][:60].ljust(60), r["Calls"].rjust(6), "%.1f us" % (float(r["AverageNs"]) / 1e3), r["Percentage"], "%")
PY
