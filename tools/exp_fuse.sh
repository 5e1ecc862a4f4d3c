#!/bin/bash
# same-box A/B: r^2 + padding written by k_nl_build (fused) vs the separate k_dd_scalars pass
for v in fused separate fused separate; do
  if [ $v = separate ]; then export POLAR_NO_FUSE_R2=1; else unset POLAR_NO_FUSE_R2; fi
  timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/fuse_$v.log 2>&1
  echo "$v"; python tools/show_line.py gpurun_out/fuse_$v.log
done
