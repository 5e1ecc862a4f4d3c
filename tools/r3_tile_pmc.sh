#!/bin/bash
# round 3 lab: kernel durations and PMC counters of the tile sweep on the configs[2] box (tools/sweep_ab.py as the driver)
tag=${1:-r3tile}
R=$PWD
export LAB_CASES="${LAB_CASES:-5x5x4:prec}" LAB_KERNELS="${LAB_KERNELS:-tile=POLAR_SWEEP_KERNEL=4}" LAB_STEPS=2
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_trace -- python $R/tools/sweep_ab.py > $R/gpurun_out/${tag}_trace.log 2>&1
echo "trace rc=$?"
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INSTS_SALU" \
           "GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/${tag}_pmc$i -- python $R/tools/sweep_ab.py > $R/gpurun_out/${tag}_pmc$i.log 2>&1
  echo "pass $i rc=$?"
done
cd $R
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/${tag}_trace/*/*kernel_stats.csv")):
    for k, r in enumerate(csv.DictReader(open(f))):
        if k < 14: print(r["Name"][:70], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])
for f in sorted(glob.glob("gpurun_out/${tag}_pmc*/*/*counter_collection.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_field_tile" in r["Kernel_Name"] or "k_tile_build" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"][:24], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(f"{k[0]:26s} {k[1]:28s} n={len(v):5d} mean={sum(v)/len(v):16.1f}")
PY
