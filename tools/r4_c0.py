#!/usr/bin/env python3
"""round 4: BASELINE configs[0] (exact mode, 1349 atoms) alone, for rocprofv3: 10 resident steps"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module(bench.PKG); wl = importlib.import_module(bench.PKG + ".workload")
r = bench.config0_exact(torch, pkg, wl, steps=10, warmup=2)
print("config0: ms/step %.3f  us/iteration %.1f  iterations %d  E_pol %.12f" % (r["ms_per_step"], r["us_per_iteration"], r["iterations"], r["eng_pol"]))
