#!/bin/bash
# round 4: region pipeline A/B on the headline (same box), then the GPU suite and the full bench line
mkdir -p gpurun_out
for rep in 1 2; do
  for pl in 0 1; do
    POLAR_PIPELINE=$pl timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=d['config']
print('pipeline $pl rep $rep: ms/step %.3f  ms_solve %.3f  sweeps %d  ms/launch %.2f us  alone %.2f us  frac %.4f  E_pol %.9f' % (d['ms_per_step'], c['ms_solve'], c['sweeps'], 1e3*d['roofline']['ms_per_launch'], 1e3*d['roofline']['alone']['ms_per_launch'], d['roofline']['frac'], c['eng_pol']))" | tee -a gpurun_out/r4f_pipeline_ab.txt
  done
done
for pl in 0 1; do
  POLAR_PIPELINE=$pl timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 2 --config 4 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=d['config']
print('config4 529k pipeline $pl: ms/step %.3f  ms_solve %.3f  sweeps %d' % (d['ms_per_step'], c['ms_solve'], c['sweeps']))" | tee -a gpurun_out/r4f_pipeline_ab.txt
  POLAR_PIPELINE=$pl timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 2 --config 1 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=d['config']
print('config1 36k pipeline $pl: ms/step %.3f  ms_solve %.3f  sweeps %d' % (d['ms_per_step'], c['ms_solve'], c['sweeps']))" | tee -a gpurun_out/r4f_pipeline_ab.txt
done
timeout -k 10 1000 python -m pytest tests -q -m gpu > gpurun_out/r4f_tests.log 2>&1
echo "tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r4f_tests.log | tail -20
timeout -k 10 900 python bench.py > gpurun_out/r4f_bench.json 2> gpurun_out/r4f_bench.err
echo "bench rc=$?"
POLAR_FORCE_LAUNCH=1 POLAR_FORCE_DIST=1 timeout -k 10 600 python bench.py --gpus 1 --steps 5 --warmup 2 > gpurun_out/r4f_bench_dist1.json 2> gpurun_out/r4f_bench_dist1.err
echo "bench dist1 rc=$?"; tail -c 300 gpurun_out/r4f_bench_dist1.err
