#!/bin/bash
# round 4: the shared-colouring tests again, then the N = 8 emulations on BASELINE configs[4] (7x7x8 = 528,808 atoms)
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "one_colouring_shared" > gpurun_out/r4c_tests.log 2>&1
echo "tests rc=$?" | tee -a gpurun_out/r4c_tests.log
grep -E "^(FAILED|ERROR)|passed|failed|^E  " gpurun_out/r4c_tests.log | tail -20
# (1) lock-step emulation: the single handle's colouring on 8 slabs, exchange per colour phase delivered 0 / 1 / 2 phases late
LAB_REPS=7x7x8 LAB_WORLD=8 LAB_GLUE=1.6 LAB_W=0,phase0,phase1,phase2 timeout -k 10 900 python tools/lab_shards.py > gpurun_out/r4c_lab_shards_n8.txt 2>&1
echo "lab_shards rc=$?"; cat gpurun_out/r4c_lab_shards_n8.txt | tail -12
# (2) the driver itself, 8 ranks over the mock transport, the colouring built by the ranks together
hipcc -O2 -std=c++17 -fPIC -shared -o /tmp/libfake_rccl.so tests/dist_mock/fake_rccl.cpp
for sched in legacy lag0 lag1; do
  POLAR_RCCL_LIB=/tmp/libfake_rccl.so MOCK_REPS=7x7x8 MOCK_DD=12.8345 timeout -k 10 900 python tests/dist_mock/run_mock_dist.py 8 precision 1 $sched > gpurun_out/r4c_mock8_$sched.json 2> gpurun_out/r4c_mock8_$sched.err
  echo "mock8 $sched rc=$?"; python - <<PY
import json
try:
    r = json.loads([l for l in open("gpurun_out/r4c_mock8_$sched.json") if l.startswith("{")][-1])
    print("$sched", "ref sweeps", r["ref"]["sweeps"], "ranks sweeps", r["ranks"][0]["sweeps"], "colors", r["ranks"][0]["ncolors"], "mu_err", r["mu_err"], "clashes", r.get("color_clashes"), "classes", r["classes"], "E_pol", r["ranks"][0]["eng_pol"], r["ref"]["eng_pol"])
except Exception as e:
    print("no result", e)
PY
done
