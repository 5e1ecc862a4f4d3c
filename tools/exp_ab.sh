#!/bin/bash
# three unprofiled bench runs (20 steps each) for A/B comparisons
for k in 1 2 3; do
  timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/ab_$k.log 2>&1
  python tools/show_line.py gpurun_out/ab_$k.log
done
