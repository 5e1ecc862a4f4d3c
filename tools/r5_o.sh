#!/bin/bash
# round 5: host looks at the loop state where the last solve ended (look_at_state): parity subset + headline A/B is not possible in one
# library, so: the headline three times, config 0, the one-RCCL-rank schedules
tag=${1:-r5o}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_fuzz.py -q -x --tb=short > gpurun_out/${tag}_tests.log 2>&1
rc=$?; echo "tests rc=$rc"; grep -E "^(FAILED|ERROR)|passed|failed|Aborted|fault" gpurun_out/${tag}_tests.log | tail -6
if grep -q "Memory access fault" gpurun_out/${tag}_tests.log; then exit 9; fi
if [ $rc -ne 0 ]; then exit $rc; fi
for k in 1 2 3; do
  timeout -k 10 300 python bench.py --direct --steps 20 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/${tag}_h$k.json 2> gpurun_out/${tag}_h$k.err
  python tools/show_line.py gpurun_out/${tag}_h$k.json | cut -c1-200
done
timeout -k 10 300 python tools/r4_c0.py 2>&1 | grep config0
POLAR_FORCE_DIST=1 timeout -k 10 600 python bench.py --gpus 1 --steps 10 --warmup 2 --no-extras > gpurun_out/${tag}_dist1.json 2> gpurun_out/${tag}_dist1.err
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/${tag}_dist1.json") if l.startswith("{")][-1])
print({k:(round(v.get("ms_per_step",0),3), v.get("sweeps")) for k,v in d["config"]["schedules"].items()})
PY
