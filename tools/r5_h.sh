#!/bin/bash
# round 5: the persistent LJ/Coulomb kernel (k_ljcoul_pers): goldens with it forced, then the headline box with and without it
tag=${1:-r5h}
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -k "persistent_lj or hip_matches_reference_golden or device_neighbor_build" --tb=short > gpurun_out/${tag}_tests.log 2>&1
rc=$?; echo "tests rc=$rc"; grep -E "^(FAILED|ERROR)|passed|failed|Aborted|fault" gpurun_out/${tag}_tests.log | head -20
if grep -q "Memory access fault" gpurun_out/${tag}_tests.log; then exit 9; fi
if [ $rc -ne 0 ]; then exit $rc; fi
for pers in 1 0; do
  POLAR_LJ_PERS=$pers timeout -k 10 300 python bench.py --direct --steps 20 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/${tag}_pers${pers}.json 2> gpurun_out/${tag}_pers${pers}.err
  echo "POLAR_LJ_PERS=$pers rc=$?"; python tools/show_line.py gpurun_out/${tag}_pers${pers}.json 2>/dev/null | head -3
done
