#!/usr/bin/env python3
"""round 4: the MD-shaped leg with LAMMPS' list uploaded every 10th step, alone (reneighbor step, polar_set_neighbors_csr)"""
import importlib, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module(bench.PKG); wl = importlib.import_module(bench.PKG + ".workload")
s = wl.replicate_fixture(os.path.join(ROOT, "tests", "golden", "mof5_h2.npz"), 5, 5, 4, extra_args=bench.PREC11 + ["dd_cutoff", repr(bench.CUT_COUL)])
for rep in range(2):
    r = bench.md_leg(pkg, s)
    print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items() if k != "what"})
