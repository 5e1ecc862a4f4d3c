#!/bin/bash
# lab: ablations of the component-per-lane sweep (k_field_quad); results to gpurun_out/quad_lab.txt
export LAB_VARIANTS="gs:0,gs:1,gs:8,gs:2,gs:4,gs:6,jac:0,jac:1,jac:2,jac:4,jac:6"
timeout -k 10 500 python tools/sweep_lab.py > gpurun_out/quad_lab.txt 2>&1
cat gpurun_out/quad_lab.txt
