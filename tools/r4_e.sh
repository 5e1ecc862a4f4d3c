#!/bin/bash
# round 4: whole GPU suite, Anderson mixing across 8 mock ranks, the full bench line, the distributed path with one RCCL rank
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -q -m gpu > gpurun_out/r4e_tests.log 2>&1
echo "tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r4e_tests.log | tail -20
hipcc -O2 -std=c++17 -fPIC -shared -o /tmp/libfake_rccl.so tests/dist_mock/fake_rccl.cpp
POLAR_RCCL_LIB=/tmp/libfake_rccl.so MOCK_REPS=7x7x8 MOCK_DD=12.8345 timeout -k 10 600 python tests/dist_mock/run_mock_dist.py 8 precision 1 accel1 > gpurun_out/r4e_mock8_accel1.json 2> gpurun_out/r4e_mock8_accel1.err
echo "mock8 accel rc=$?"; python - <<PY
import json
try:
    r = json.loads([l for l in open("gpurun_out/r4e_mock8_accel1.json") if l.startswith("{")][-1])
    print("accel1", "ref sweeps", r["ref"]["sweeps"], "ranks sweeps", r["ranks"][0]["sweeps"], "mu_err", r["mu_err"], "E_pol", r["ranks"][0]["eng_pol"], r["ref"]["eng_pol"], "exchanges", r["ranks"][0]["exchanges"])
except Exception as e:
    print("no result", e)
PY
timeout -k 10 900 python bench.py > gpurun_out/r4e_bench.json 2> gpurun_out/r4e_bench.err
echo "bench rc=$?"; tail -c 400 gpurun_out/r4e_bench.err
POLAR_FORCE_DIST=1 timeout -k 10 600 python bench.py --gpus 1 --steps 5 --warmup 2 > gpurun_out/r4e_bench_dist1.json 2> gpurun_out/r4e_bench_dist1.err
echo "bench dist1 rc=$?"; tail -c 300 gpurun_out/r4e_bench_dist1.err
