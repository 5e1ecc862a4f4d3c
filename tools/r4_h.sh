#!/bin/bash
# round 4: config 0 with the component-major tensor (fused one-launch-per-block form against the two-launch form), then the
# profiled headline + full bench line (gpu_profile.sh) and the PMC traffic passes
mkdir -p gpurun_out
for v in fused two; do
  if [ $v = two ]; then export POLAR_GS_TWO_LAUNCH=1; else unset POLAR_GS_TWO_LAUNCH; fi
  python - <<PY | tee -a gpurun_out/r4h_config0.txt
import importlib, sys, torch
sys.path.insert(0, ".")
import bench
pkg = importlib.import_module(bench.PKG); wl = importlib.import_module(bench.PKG + ".workload")
r = bench.config0_exact(torch, pkg, wl, steps=20, warmup=3)
print("config0 $v: ms/step %.3f  us/iteration %.1f  iterations %d  E_pol %.12f" % (r["ms_per_step"], r["us_per_iteration"], r["iterations"], r["eng_pol"]))
PY
done
unset POLAR_GS_TWO_LAUNCH
bash tools/gpu_profile.sh r4h
bash tools/pmc_traffic.sh r4ht > gpurun_out/r4h_traffic.log 2>&1
tail -3 gpurun_out/r4h_traffic.log | cut -c1-200
