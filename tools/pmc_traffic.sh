#!/bin/bash
# HBM traffic of the sweep kernel: FETCH_SIZE and WRITE_SIZE in separate passes (MI355X_MICROARCH.md)
tag=${1:-traffic}
R=$PWD
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/${tag}_pmc$i -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${tag}_pmc$i.log 2>&1
  echo "pass $i rc=$?"
done
