#!/usr/bin/env python3
"""Print the bench line and the top kernels of a gpurun_out/<tag>_{bench.log,prof} pair."""
import csv, glob, json, sys
tag = sys.argv[1]
for ln in open(f"gpurun_out/{tag}_bench.log"):
    if ln.startswith('{"metric"'):
        d = json.loads(ln); c = d["config"]
        print("ms/step %.2f  value %.0f  solve %.2f  per-iter %.3f  colors %s  lj %.2f  force %.2f  list %.2f  static %.2f  frac %.3f" % (
            d["ms_per_step"], d["value"], c["ms_solve"], c["ms_per_dipole_iteration"], c["colors"], c["ms_ljcoul"],
            c["ms_force"], c["ms_list"], c["ms_static"], d["roofline"]["frac"]))
f = glob.glob(f"gpurun_out/{tag}_prof/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:int(sys.argv[2]) if len(sys.argv) > 2 else 9]:
    print(r["Name"][:58].ljust(58), r["Calls"].rjust(5), ("%.1f" % (float(r["AverageNs"]) / 1e3)).rjust(9), "us", r["Percentage"].rjust(6), "%",
          ("min %.1f max %.1f" % (float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3)))
