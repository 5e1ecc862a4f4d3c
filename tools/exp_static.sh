#!/bin/bash
# A/B of row-kernel variants with a3 kept off its side stream so that every kernel runs alone (bench headline, no extras)
# usage: bash tools/exp_static.sh VAR   (VAR=1 against VAR=0, two repetitions), e.g. POLAR_STATIC_XQ, POLAR_LJ_TYPED
var=${1:-POLAR_STATIC_XQ}
run() { env "$@" POLAR_NO_OVERLAP=1 timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('$*'.ljust(24), round(d['ms_per_step'],3), 'lj', round(c['ms_ljcoul'],3), 'static', round(c['ms_static'],3), 'solve', round(c['ms_solve'],3), 'E', c['eng_pol'])
"; }
for rep in 1 2; do run $var=1; run $var=0; done
