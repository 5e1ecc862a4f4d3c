#!/bin/bash
# standalone k_nl_build time with / without the per-atom stencil trimming (a3 kept off the side stream)
export POLAR_NO_OVERLAP=1
R=$PWD
for v in 1 0; do
  export POLAR_NL_TRIM=$v
  (cd /tmp && TMPDIR=/tmp timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/trim${v}_prof -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/trim${v}.log 2>&1)
  echo "trim=$v"; python tools/show_line.py gpurun_out/trim${v}.log
  python - <<PY
import csv, glob
f = glob.glob("gpurun_out/trim${v}_prof/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "k_nl_build" in r["Name"]: print("  k_nl_build avg %.1f us" % (float(r["AverageNs"]) / 1e3))
PY
done
