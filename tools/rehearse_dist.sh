#!/bin/bash
# Rehearsal of `bench.py --gpus N` with world > 1 on a ONE-GPU box: N processes share the GPU, torch.distributed runs on
# gloo with every exchange staged through the host (parallel.HostStagedDist).  Checks the glue of bench_distributed --
# shard construction, point-to-point halo plans, device-built lists of the own rows, the all-reduced stop rule, the JSON
# line -- against the single-process run of the same box (E_pol, sweeps).  Timings of these runs mean nothing.
# usage: bash tools/rehearse_dist.sh <tag> [N ...]      (N <= 5: the pool allows 6 processes on a card and the launcher is one of them)
set -u
tag=${1:-reh}; shift
ns=${@:-2 4}
out=gpurun_out
mkdir -p $out
export POLAR_DIST_BACKEND=gloo
port=29517
for n in $ns; do
  cfg=3; [ "$n" -ge 5 ] && cfg=4
  timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $port \
      bench.py --gpus $n --steps 2 --warmup 1 --config $cfg > $out/${tag}_n$n.log 2>&1
  echo "N=$n config $cfg rc=$?"
  grep '"metric"' $out/${tag}_n$n.log | python -c "
import json,sys
for ln in sys.stdin:
    d=json.loads(ln); c=d['config']
    print('  natoms', c['natoms'], 'sweeps', c['sweeps'], 'E_pol', c['eng_pol'], 'rms', c['rms_dmu_last_sweep'], 'held0', c['atoms_held_rank0'], 'ms/step', round(d['ms_per_step'],2))
"
  port=$((port+1))
done
unset POLAR_DIST_BACKEND
for cfg in ${REH_SINGLE:-3}; do
  timeout -k 10 300 python bench.py --config $cfg --steps 2 --warmup 1 --no-extras --no-cpu-baseline > $out/${tag}_single_c$cfg.log 2>&1
  echo "single config $cfg rc=$?"
  grep '"metric"' $out/${tag}_single_c$cfg.log | python -c "
import json,sys
for ln in sys.stdin:
    d=json.loads(ln); c=d['config']
    print('  natoms', c['natoms'], 'sweeps', c['sweeps'], 'E_pol', c['eng_pol'], 'rms', c['rms_dmu_last_sweep'], 'ms/step', round(d['ms_per_step'],2))
"
done
