#!/bin/bash
for v in 1 2 1 2; do
  POLAR_CACHE_R2=$v timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/nc_$v.log 2>&1
  echo "cache_r2=$v"; python tools/show_line.py gpurun_out/nc_$v.log
done
POLAR_CACHE_R2=2 timeout -k 10 400 python bench.py --reps 5 5 4 --steps 3 --warmup 1 --no-cpu-baseline --extra fixed_iteration no precision 1e-11 max_iterations 100 > gpurun_out/nc_c2.log 2>&1
echo "config2 cache_r2=2"; python tools/show_line.py gpurun_out/nc_c2.log
POLAR_CACHE_R2=1 timeout -k 10 400 python bench.py --reps 5 5 4 --steps 3 --warmup 1 --no-cpu-baseline --extra fixed_iteration no precision 1e-11 max_iterations 100 > gpurun_out/nc_c1.log 2>&1
echo "config2 cache_r2=1"; python tools/show_line.py gpurun_out/nc_c1.log
