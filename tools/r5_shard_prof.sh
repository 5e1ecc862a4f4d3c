#!/bin/bash
tag=${1:-r5s}
R=$PWD
cd /tmp && export TMPDIR=/tmp
for a in 0 4; do
  LAB_ACCEL=$a timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_a${a}_prof -- python $R/tools/r5_shard_prof.py > $R/gpurun_out/${tag}_a$a.log 2>&1
  echo "accel $a rc=$?"; grep -E "^rank|^\{" $R/gpurun_out/${tag}_a$a.log | cut -c1-300
  python - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/${tag}_a${a}_prof/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print("  ", r["Name"][:60].ljust(60), r["Calls"].rjust(5), ("%.1f" % (float(r["AverageNs"]) / 1e3)).rjust(8), "us", r["Percentage"].rjust(6), "%")
PY
done
