#!/bin/bash
# HBM-side traffic of the sweep kernel on the bench headline (BASELINE configs[2]): FETCH_SIZE and WRITE_SIZE in separate
# --pmc passes (MI355X_MICROARCH.md, HBM section), then tools/traffic_json.py turns the per-launch means into
# profiles/traffic_pmc.json, which bench.py reads for roofline.traffic (keyed by kernel version + atom count).
tag=${1:-traffic}
R=$PWD
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/${tag}_pmc$i -- python $R/bench.py --direct --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/${tag}_pmc$i.log 2>&1
  echo "pass $i ($grp) rc=$?"
done
cd $R && python tools/traffic_json.py $tag
