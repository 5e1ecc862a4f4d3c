#!/bin/bash
for v in 0 1 0 1; do
  POLAR_PACK_JR=$v timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/pjr_$v.log 2>&1
  echo "pack_jr=$v"; python tools/show_line.py gpurun_out/pjr_$v.log
done
