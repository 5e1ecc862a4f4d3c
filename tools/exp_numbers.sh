#!/bin/bash
# collect the numbers DESIGN.md quotes: config 0 timing, config 2 (135k, precision 1e-11), full default bench
timeout -k 10 300 python tools/config0_timing.py --cpu > gpurun_out/config0_timing.txt 2>&1
tail -6 gpurun_out/config0_timing.txt
timeout -k 10 400 python bench.py --reps 5 5 4 --steps 3 --warmup 1 --no-cpu-baseline --extra fixed_iteration no precision 1e-11 max_iterations 100 > gpurun_out/config2.log 2>&1
grep '"metric"' gpurun_out/config2.log | cut -c1-300
timeout -k 10 500 python bench.py > gpurun_out/bench_full.log 2>&1
grep '"metric"' gpurun_out/bench_full.log
