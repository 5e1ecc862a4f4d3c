#!/bin/bash
# round 5: the driver itself on 8 mock ranks, BASELINE configs[4] (7x7x8), the LEGACY schedule with polar_accel 4 (what
# `bench.py --gpus 8 --schedule legacy_accel4` runs) -- sweep count and the per-part profile of rank 0
tag=${1:-r5j}
mkdir -p gpurun_out
/opt/rocm/bin/hipcc -O2 -std=c++17 -fPIC -shared -o /tmp/libfake_rccl.so tests/dist_mock/fake_rccl.cpp
for sched in accelL legacy; do
  POLAR_RCCL_LIB=/tmp/libfake_rccl.so MOCK_REPS=7x7x8 MOCK_DD=12.8345 timeout -k 10 900 python tests/dist_mock/run_mock_dist.py 8 precision 1 $sched > gpurun_out/${tag}_mock8_$sched.json 2> gpurun_out/${tag}_mock8_$sched.err
  echo "$sched rc=$?"
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/${tag}_mock8_$sched.json") if l.startswith("{")][-1])
r=d["ranks"][0]
print("$sched: single handle", d["ref"]["sweeps"], "sweeps; 8 ranks", r["sweeps"], "sweeps, exchanges", r["exchanges"], "all-reduces", r["allreduces"], "mu_err %.2e" % d["mu_err"], "E_pol", r["eng_pol"], "ref", d["ref"]["eng_pol"])
print("   rank 0 profile (ms, 8 threads sharing ONE GPU: shares only):", {k: round(v,3) for k,v in r["profile"].items()}, "ms_solve", round(r["ms_solve"],3))
PY
done
