#!/bin/bash
# round 5: k_nl_build FAST (wrapped pos4, images from cell offsets): parity (everything in list mode), then the headline with it on / off
tag=${1:-r5q}
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -q -m gpu -x --tb=short > gpurun_out/${tag}_tests.log 2>&1
rc=$?; echo "tests rc=$rc"; grep -E "^(FAILED|ERROR)|passed|failed|Aborted|fault" gpurun_out/${tag}_tests.log | tail -6
if grep -q "Memory access fault" gpurun_out/${tag}_tests.log; then exit 9; fi
if [ $rc -ne 0 ]; then tail -40 gpurun_out/${tag}_tests.log; exit $rc; fi
for f in 1 0 1 0; do
  POLAR_NL_FAST=$f timeout -k 10 300 python bench.py --direct --steps 20 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/${tag}_f$f.json 2> gpurun_out/${tag}_f$f.err
  echo -n "POLAR_NL_FAST=$f: "; python tools/show_line.py gpurun_out/${tag}_f$f.json | cut -c1-200
done
