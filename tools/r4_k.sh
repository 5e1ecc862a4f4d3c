#!/bin/bash
mkdir -p gpurun_out
POLAR_DIST_NOSPLIT=1 timeout -k 10 500 python tools/r4_dist_cost.py 2 > gpurun_out/r4k_dist_cost_nosplit.txt 2>&1
echo "rc=$?"; grep -E "^(plain|legacy|lag)" gpurun_out/r4k_dist_cost_nosplit.txt
