#!/bin/bash
# round 5: k_gs_blk with 16-byte loads in its push: exact-mode parity tests, then timing and the kernel's rocprofv3 record
tag=${1:-r5m}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_edges.py -q -k "exact or golden or knife or block_inverses or bit_reproducible or small_random or triclinic" --tb=short > gpurun_out/${tag}_tests.log 2>&1
rc=$?; echo "tests rc=$rc"; grep -E "^(FAILED|ERROR)|passed|failed|Aborted|fault" gpurun_out/${tag}_tests.log | tail -8
if grep -q "Memory access fault" gpurun_out/${tag}_tests.log; then exit 9; fi
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python tools/r4_c0.py 2>&1 | grep config0
timeout -k 10 600 python tools/r4_x10k.py 2>&1 | tail -1 | cut -c1-700
bash tools/r5_a.sh ${tag} > gpurun_out/${tag}_prof.log 2>&1; tail -3 gpurun_out/${tag}_prof.log
