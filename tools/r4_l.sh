#!/bin/bash
mkdir -p gpurun_out
LAB_REPS=7x7x8 LAB_WORLD=8 LAB_GLUE=1.6 LAB_W=phase1e2,phase1e4,phase0e2,phase0e4,phase2e2 timeout -k 10 900 python tools/lab_shards.py > gpurun_out/r4l_lab_shards_every.txt 2>&1
echo "rc=$?"; grep -v amdgpu.ids gpurun_out/r4l_lab_shards_every.txt | tail -8
