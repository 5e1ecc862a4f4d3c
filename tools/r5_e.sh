#!/bin/bash
# round 5: the rebuilt bench.py paths on the 1-GPU box: dist tests on the mock transport, then the launcher with one RCCL rank (all
# three schedules as child jobs), the gloo rehearsals with 2 and 4 ranks sharing the GPU, and the N = 1 line through the launcher
tag=${1:-r5e}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -k "in_library" --tb=short > gpurun_out/${tag}_dist_tests.log 2>&1
echo "dist tests rc=$?"; tail -3 gpurun_out/${tag}_dist_tests.log | cut -c1-300
SECONDS=0
POLAR_FORCE_DIST=1 timeout -k 10 900 python bench.py --steps 5 --warmup 1 > gpurun_out/${tag}_dist1.json 2> gpurun_out/${tag}_dist1.err
echo "one RCCL rank rc=$? wall ${SECONDS}s"; tail -1 gpurun_out/${tag}_dist1.json | cut -c1-300
SECONDS=0
POLAR_DIST_BACKEND=gloo timeout -k 10 900 python bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/${tag}_gloo2.json 2> gpurun_out/${tag}_gloo2.err
echo "gloo 2 rc=$? wall ${SECONDS}s"; tail -1 gpurun_out/${tag}_gloo2.json | cut -c1-300
SECONDS=0
timeout -k 10 900 python bench.py --steps 5 --warmup 1 > gpurun_out/${tag}_n1.json 2> gpurun_out/${tag}_n1.err
echo "N=1 rc=$? wall ${SECONDS}s"; tail -1 gpurun_out/${tag}_n1.json | cut -c1-300
