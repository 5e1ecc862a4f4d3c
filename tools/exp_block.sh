#!/bin/bash
for v in 256 64 128 512 1024 256; do
  POLAR_QUAD_BLOCK=$v timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/qb_$v.log 2>&1
  echo "quad_block=$v"; python tools/show_line.py gpurun_out/qb_$v.log
done
