#!/bin/bash
# round 5: threads of a3's persistent workgroup: how much of the CU it leaves to the list build beside it
tag=${1:-r5l}
for t in 1024 768 512 384 256; do
  POLAR_LJ_PERS_THREADS=$t timeout -k 10 300 python bench.py --direct --steps 20 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/${tag}_t$t.json 2> gpurun_out/${tag}_t$t.err
  echo -n "threads $t: "; python tools/show_line.py gpurun_out/${tag}_t$t.json | cut -c1-200
done
