#!/usr/bin/env python3
"""round 4: BASELINE configs[0] through the LAB library with POLAR_GS_STAMPS=1: where workgroup 0 of the per-block launch spends its time"""
import importlib, os, sys
os.environ["POLAR_GS_STAMPS"] = "1"
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module(bench.PKG); wl = importlib.import_module(bench.PKG + ".workload")
s, _ = wl.load_fixture(os.path.join(ROOT, "tests", "golden", "mof5_h2.npz"),
                       extra_args=["use_previous", "no", "polar_gs_ranked", "yes", "precision", "1e-11", "max_iterations", "30"])
p = pkg.pair_from_system(s, lab=True)
for k in range(3):
    out = p.compute()
print("iterations", out["iterations"], "E_pol", out["eng_pol"])
