#!/bin/bash
# round 5: rocprofv3 record of exact mode at 10,792 atoms -- kernel stats, then FETCH_SIZE / WRITE_SIZE of
# k_gs_blk in separate --pmc passes (MI355X_MICROARCH.md, HBM section); the same for config 0 (1,349 atoms) kernel stats
tag=${1:-r5a}
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_x10k_prof -- python $R/tools/r4_x10k.py > $R/gpurun_out/${tag}_x10k.log 2>&1
echo "x10k stats rc=$?"; tail -1 $R/gpurun_out/${tag}_x10k.log | cut -c1-400
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/${tag}_x10k_pmc$i -- python $R/tools/r4_x10k.py > $R/gpurun_out/${tag}_x10k_pmc$i.log 2>&1
  echo "pass $i ($grp) rc=$?"
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_c0_prof -- python $R/tools/r4_c0.py > $R/gpurun_out/${tag}_c0.log 2>&1
echo "c0 stats rc=$?"
