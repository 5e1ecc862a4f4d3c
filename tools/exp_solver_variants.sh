export POLAR_DEBUG=1
for cfg in "jacobi:--extra polar_gs_ranked no" "gs26:" ; do
  name=${cfg%%:*}; args=${cfg#*:}
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline $args > gpurun_out/exp_$name.log 2>&1
done
POLAR_COLOR_DIST=1.2 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/exp_gs12.log 2>&1
POLAR_COLOR_DIST=4.0 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/exp_gs40.log 2>&1
grep -h "colour\|metric" gpurun_out/exp_*.log | cut -c1-400
