#!/usr/bin/env python3
"""round 3 lab: bench.py's ballistic MD leg alone (for rocprofv3: where a colour rebuild spends its time)"""
import importlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
pkg = importlib.import_module(bench.PKG); wl = importlib.import_module(bench.PKG + ".workload")
cfg = bench.CONFIGS[2]
s = bench.build_workload(wl, cfg["reps"], solver=cfg["solver"])
r = bench.md_leg(pkg, s, steps=int(os.environ.get("LAB_STEPS", "60")), device_neigh=True, motion="ballistic", temperature=float(os.environ.get("LAB_T", "300")))
r.pop("what")
print(json.dumps(r))
