#!/bin/bash
# round 5, final state: GPU suite, profiled headline + full bench line (through the launcher), PMC traffic passes, the distributed
# path (one RCCL rank through the launcher: three schedules; two ranks over gloo), the refusal of --gpus 2 on one GPU, smoke()
tag=${1:-r5z}
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -q -m gpu > gpurun_out/${tag}_tests.log 2>&1
rc=$?; echo "tests rc=$rc"; grep -E "^(FAILED|ERROR)|passed|failed|Aborted|fault" gpurun_out/${tag}_tests.log | tail -10
if grep -q "Memory access fault" gpurun_out/${tag}_tests.log; then exit 9; fi
bash tools/gpu_profile.sh ${tag}
bash tools/pmc_traffic.sh ${tag}t > gpurun_out/${tag}_traffic.log 2>&1
tail -2 gpurun_out/${tag}_traffic.log | cut -c1-160
POLAR_FORCE_DIST=1 timeout -k 10 600 python bench.py --gpus 1 --steps 10 --warmup 2 > gpurun_out/${tag}_bench_dist1.json 2> gpurun_out/${tag}_bench_dist1.err
echo "bench dist1 rc=$?"
POLAR_DIST_BACKEND=gloo timeout -k 10 600 python3 bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/${tag}_gpus2_gloo.json 2> gpurun_out/${tag}_gpus2_gloo.err
echo "gpus2 gloo rc=$?"
python3 bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/${tag}_gpus2_nccl.log 2>&1
echo "gpus2 nccl rc=$? (expected 2 on a 1-GPU box)"; cat gpurun_out/${tag}_gpus2_nccl.log | tail -2
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${tag}_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/${tag}_smoke.log
exit $rc
