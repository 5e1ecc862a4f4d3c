#!/bin/bash
# round 3: GPU suite, profiled headline, full bench line, PMC traffic / issue passes; results under gpurun_out/<tag>_*
tag=${1:-r3p}
bash tools/r3_gpu_tests.sh ${tag}
bash tools/gpu_profile.sh ${tag}
bash tools/pmc_traffic.sh ${tag}t > gpurun_out/${tag}_traffic.log 2>&1
tail -3 gpurun_out/${tag}_traffic.log | cut -c1-200
