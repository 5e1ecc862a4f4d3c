#!/bin/bash
# lab: cached (s3,s5) vs cached r^2 in the quad sweep
for v in 0 1; do
  for mode in "" "--extra polar_gs_ranked no"; do
    POLAR_CACHE_R2=$v timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline $mode > gpurun_out/r2_$v.log 2>&1
    python - <<PY
import json
for ln in open("gpurun_out/r2_$v.log"):
    if ln.startswith('{"metric"'):
        d = json.loads(ln); c = d["config"]
        print("cache_r2=$v mode='$mode' ms/step %.3f solve %.3f per-iter %.4f colors %s eng_pol %.10f" % (d["ms_per_step"], c["ms_solve"], c["ms_per_dipole_iteration"], c["colors"], c["eng_pol"]))
PY
  done
done
