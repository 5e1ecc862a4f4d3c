#!/bin/bash
# standalone kernel times (a3 kept off the side stream): rocprofv3 kernel stats of the default bench
export POLAR_NO_OVERLAP=1
R=$PWD
(cd /tmp && TMPDIR=/tmp timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/sa_prof -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/sa.log 2>&1)
python tools/show_line.py gpurun_out/sa.log
python - <<PY
import csv, glob
f = sorted(glob.glob("gpurun_out/sa_prof/*/*kernel_stats.csv"))[-1]
for r in list(csv.DictReader(open(f)))[:9]:
    print("  %-52s %5s calls  avg %8.1f us" % (r["Name"][:52], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
