#!/bin/bash
# round 4, final state: GPU suite, profiled headline + full bench line, PMC traffic, the distributed path (one RCCL rank; two ranks over gloo)
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -q -m gpu > gpurun_out/r4f6_tests.log 2>&1
echo "tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|Aborted" gpurun_out/r4f6_tests.log | tail -10
bash tools/gpu_profile.sh r4f6
bash tools/pmc_traffic.sh r4f6t > gpurun_out/r4f6_traffic.log 2>&1
tail -2 gpurun_out/r4f6_traffic.log | cut -c1-160
POLAR_FORCE_LAUNCH=1 POLAR_FORCE_DIST=1 timeout -k 10 600 python bench.py --gpus 1 --steps 5 --warmup 2 > gpurun_out/r4f6_bench_dist1.json 2> gpurun_out/r4f6_bench_dist1.err
echo "bench dist1 rc=$?"
POLAR_DIST_BACKEND=gloo timeout -k 10 600 python3 bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/r4f6_gpus2_gloo.json 2> gpurun_out/r4f6_gpus2_gloo.err
echo "gpus2 gloo rc=$?"
python3 bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/r4f6_gpus2_nccl.log 2>&1
echo "gpus2 nccl rc=$? (expected 2 on a 1-GPU box)"; cat gpurun_out/r4f6_gpus2_nccl.log | tail -2
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r4f6_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r4f6_smoke.log
