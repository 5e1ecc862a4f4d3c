#!/bin/bash
# round 4: BASELINE configs[0] under rocprofv3 (the exact-order Gauss-Seidel chain), then the GPU suite
mkdir -p gpurun_out
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4i_c0 -- python $R/tools/r4_c0.py > $R/gpurun_out/r4i_c0.log 2>&1
echo "rc=$?"; grep config0 $R/gpurun_out/r4i_c0.log
python - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/r4i_c0/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:6]:
    print("  %-60s calls %6s avg %9.1f ns  total %8.3f ms  %5s %%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]), float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
cd $R
python tools/r4_c0.py | grep config0
timeout -k 10 1000 python -m pytest tests -q -m gpu > gpurun_out/r4i_tests.log 2>&1
echo "tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|Aborted" gpurun_out/r4i_tests.log | tail -20
