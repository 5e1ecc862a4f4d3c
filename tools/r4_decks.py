#!/usr/bin/env python3
"""round 4: the reference's three example decks in exact mode (reference semantics), resident steps: ms per step"""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module(bench.PKG); wl = importlib.import_module(bench.PKG + ".workload")
for name in ("bulk_h2", "mof5_methane", "mof5_h2"):
    path = os.path.join(ROOT, "tests", "golden", name + ".npz")
    if not os.path.exists(path):
        continue
    s, _ = wl.load_fixture(path, extra_args=["use_previous", "no", "polar_gs_ranked", "yes", "precision", "1e-11", "max_iterations", "100"])
    p = pkg.pair_from_system(s)
    out, dt, ms_solve, _ = bench.timed_steps(torch, p, 10, 2)
    p.close()
    print(f"{name:14s} {s.nlocal:5d} atoms: {1e3 * dt / 10:7.3f} ms/step, {out['iterations']} iterations, E_pol {out['eng_pol']:.9f}", flush=True)
