#!/bin/bash
# round 5: the whole GPU suite ONCE with POLAR_POISON=1 (every new device buffer filled with 0x7F bytes: a read-before-write shows
# at once), --capture=sys so that a message of the HSA runtime on fd 2 survives; then config 0 and the 10,792-atom replica bare
tag=${1:-r5b}
mkdir -p gpurun_out
POLAR_POISON=1 timeout -k 10 1000 python -m pytest tests -q -m gpu --capture=sys > gpurun_out/${tag}_tests.log 2>&1
echo "tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|Aborted|fault" gpurun_out/${tag}_tests.log | tail -15
timeout -k 10 300 python tools/r4_c0.py > gpurun_out/${tag}_c0.txt 2>&1 && grep config0 gpurun_out/${tag}_c0.txt
timeout -k 10 600 python tools/r4_x10k.py > gpurun_out/${tag}_x10k.txt 2>&1 && tail -1 gpurun_out/${tag}_x10k.txt | cut -c1-500
