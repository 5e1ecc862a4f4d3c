#!/bin/bash
# round 3: bench.py's distributed path with ONE rank on the one GPU (POLAR_FORCE_DIST=1): the C++ RCCL driver end to end
POLAR_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout -k 10 600 python bench.py --steps 5 --warmup 2 --config 2 > gpurun_out/r3_dist2.log 2>&1
tail -3 gpurun_out/r3_dist2.log | cut -c1-1800
POLAR_DIST_DRIVER=python POLAR_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29534 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout -k 10 600 python bench.py --steps 5 --warmup 2 --config 2 > gpurun_out/r3_dist2py.log 2>&1
tail -1 gpurun_out/r3_dist2py.log | cut -c1-400
