#!/bin/bash
# round 4, first contact: GPU suite on the split library, then bench.py as its own launcher
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4a_tests.log 2>&1
echo "tests rc=$?" | tee -a gpurun_out/r4a_tests.log
python3 bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/r4a_gpus2_nccl.log 2>&1
echo "gpus2 nccl rc=$? (expected non-zero on a 1-GPU box)" | tee -a gpurun_out/r4a_gpus2_nccl.log
POLAR_DIST_BACKEND=gloo timeout -k 10 600 python3 bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/r4a_gpus2_gloo.log 2> gpurun_out/r4a_gpus2_gloo.err
echo "gpus2 gloo rc=$?" | tee -a gpurun_out/r4a_gpus2_gloo.log
tail -c 600 gpurun_out/r4a_tests.log
