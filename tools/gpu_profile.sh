#!/bin/bash
# rocprofv3 kernel stats of the bench headline + the full default bench line; results under gpurun_out/<tag>_*
tag=${1:-prof}
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_prof -- python $R/bench.py --direct --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $R/gpurun_out/${tag}_bench_headline.log 2>&1
echo "profiled rc=$?"
cd $R
SECONDS=0; timeout -k 10 900 python bench.py > gpurun_out/${tag}_bench_full.log 2> gpurun_out/${tag}_bench_full.err; echo "bench wall ${SECONDS} s"
echo "full rc=$?"
tail -1 gpurun_out/${tag}_bench_full.log | cut -c1-300
