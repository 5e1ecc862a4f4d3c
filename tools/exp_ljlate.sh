#!/bin/bash
for v in 0 1; do
  if [ $v = 1 ]; then export POLAR_LJ_LATE=1; fi
  for k in 1 2; do
    timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/ljl_$v$k.log 2>&1
    echo "late=$v"; python tools/show_line.py gpurun_out/ljl_$v$k.log
  done
done
