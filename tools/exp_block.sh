for fb in 64 128 256 512 1024; do
  for mode in "jac:--extra polar_gs_ranked no" "gs:"; do
    name=${mode%%:*}; args=${mode#*:}
    POLAR_FIELD_BLOCK=$fb python bench.py --steps 5 --warmup 2 --no-cpu-baseline $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; print('fb=$fb $name ms/step %.2f solve %.2f per-iter %.3f'%(d['ms_per_step'], c['ms_solve'], c['ms_per_dipole_iteration']))" | tee -a gpurun_out/exp_block.log
  done
done
