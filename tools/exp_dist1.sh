#!/bin/bash
# rehearse the distributed bench path with one rank (RCCL world of 1) beside the single-GPU path
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
POLAR_FORCE_DIST=1 timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/dist1.log 2>&1
grep '"metric"' gpurun_out/dist1.log | cut -c1-260
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/single.log 2>&1
grep '"metric"' gpurun_out/single.log | cut -c1-200
