#!/bin/bash
# round 5: whole GPU suite on the refactored library, then the launcher rehearsals that are still missing: 4 gloo ranks sharing the
# GPU, and the FOREIGN-launcher path (torch.distributed.run started by somebody else: one line, in-process schedules, watchdog)
tag=${1:-r5f}
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -q -m gpu --tb=short > gpurun_out/${tag}_tests.log 2>&1
rc=$?; echo "suite rc=$rc"; grep -E "^(FAILED|ERROR)|passed|failed|Aborted|fault" gpurun_out/${tag}_tests.log | head -20
if grep -q "Memory access fault" gpurun_out/${tag}_tests.log; then exit 9; fi
SECONDS=0
POLAR_DIST_BACKEND=gloo timeout -k 10 900 python bench.py --gpus 4 --steps 2 --warmup 1 > gpurun_out/${tag}_gloo4.json 2> gpurun_out/${tag}_gloo4.err
echo "gloo 4 rc=$? wall ${SECONDS}s"; tail -1 gpurun_out/${tag}_gloo4.json | cut -c1-200
SECONDS=0
POLAR_FORCE_DIST=1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 5 --warmup 1 > gpurun_out/${tag}_foreign1.json 2> gpurun_out/${tag}_foreign1.err
echo "foreign launcher, one RCCL rank rc=$? wall ${SECONDS}s lines=$(grep -c '^{' gpurun_out/${tag}_foreign1.json)"; tail -1 gpurun_out/${tag}_foreign1.json | cut -c1-200
SECONDS=0
POLAR_BENCH_EXTRAS_BUDGET=3 POLAR_FORCE_DIST=1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 1 --steps 5 --warmup 1 > gpurun_out/${tag}_foreign1_wd.json 2> gpurun_out/${tag}_foreign1_wd.err
echo "the same with a 3-s extras budget (watchdog) rc=$? wall ${SECONDS}s lines=$(grep -c '^{' gpurun_out/${tag}_foreign1_wd.json)"; tail -1 gpurun_out/${tag}_foreign1_wd.json | grep -o '"extras": "[^"]*"'
exit $rc
