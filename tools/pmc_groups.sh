#!/bin/bash
# PMC passes over bench.py, one run per counter group (rocprofv3 refuses mixed trace domains with --pmc).
# usage: tools/pmc_groups.sh <tag> "<group1>|<group2>|..." [bench args]; CSV under gpurun_out/<tag>_pmcN
tag=$1; groups=$2; shift 2
R=$PWD
cd /tmp && export TMPDIR=/tmp
i=0
IFS='|' read -ra G <<< "$groups"
for grp in "${G[@]}"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/${tag}_pmc$i -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $R/gpurun_out/${tag}_pmc$i.log 2>&1
  echo "pass $i ($grp) rc=$?"
done
