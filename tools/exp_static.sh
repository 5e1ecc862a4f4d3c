#!/bin/bash
# A/B of the static-field gather (32-byte {x,y,z,q} records against whole AtomRecs), a3 kept off its side stream so the kernels run alone
run() { env "$@" POLAR_NO_OVERLAP=1 timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('$*'.ljust(24), round(d['ms_per_step'],3), 'static', round(c['ms_static'],3), 'solve', round(c['ms_solve'],3), 'E_pol', c['eng_pol'])
"; }
for rep in 1 2; do run POLAR_STATIC_XQ=1; run POLAR_STATIC_XQ=0; done
