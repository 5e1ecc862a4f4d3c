#!/bin/bash
mkdir -p gpurun_out
R=$PWD
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py -q -m gpu -x -k "golden or exact or tiny or small_random or debug or knife or matrix_free or empty or no_polarizable" > gpurun_out/r4j_tests.log 2>&1
rc=$?; echo "tests rc=$rc"; grep -E "^(FAILED|ERROR)|passed|failed|Aborted|^E  " gpurun_out/r4j_tests.log | tail
if [ $rc -ne 0 ]; then exit 0; fi
for v in wide two; do
  if [ $v = two ]; then export POLAR_GS_TWO_LAUNCH=1; else unset POLAR_GS_TWO_LAUNCH; fi
  python tools/r4_c0.py | grep config0 | sed "s/^/$v: /"
done
unset POLAR_GS_TWO_LAUNCH
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4j_c0 -- python $R/tools/r4_c0.py > $R/gpurun_out/r4j_c0.log 2>&1
grep config0 $R/gpurun_out/r4j_c0.log
python - <<PY
import csv, glob
f = sorted(glob.glob("$R/gpurun_out/r4j_c0/*/*kernel_stats.csv"))[-1]
for r in list(csv.DictReader(open(f)))[:3]:
    print("  %-60s calls %6s avg %9.1f ns  total %8.3f ms  %5s %%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]), float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
