#!/usr/bin/env python3
"""Average PMC counter values per kernel from gpurun_out/<tag>_pmc*/ (rocprofv3 counter_collection.csv)."""
import csv, glob, sys, collections
tag, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "k_field")
for f in sorted(glob.glob(f"gpurun_out/{tag}_pmc*/*/*counter_collection.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(f"{k:34s} n={len(v):5d} mean={sum(v)/len(v):16.1f}")
