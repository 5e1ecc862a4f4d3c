#!/bin/bash
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "rccl_driver" > gpurun_out/r3_dist1_tests.log 2>&1; tail -15 gpurun_out/r3_dist1_tests.log | cut -c1-250
