#!/bin/bash
timeout -k 10 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/auto_d.log 2>&1
python tools/show_line.py gpurun_out/auto_d.log
timeout -k 10 400 python bench.py --reps 5 5 4 --steps 3 --warmup 1 --no-cpu-baseline --extra fixed_iteration no precision 1e-11 max_iterations 100 > gpurun_out/config2.log 2>&1
python tools/show_line.py gpurun_out/config2.log
