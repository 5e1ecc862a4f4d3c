#!/usr/bin/env python3
"""Lab: build the bench workload ONCE, then time the solve under several env-var variants
(POLAR_ABLATE bits, POLAR_FIELD_BLOCK, solver flavour).  Scratch tool, results to stdout."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module(bench.PKG); wl = importlib.import_module(bench.PKG + ".workload")
import torch
torch.cuda.set_device(0)
reps = tuple(int(v) for v in os.environ.get("LAB_REPS", "3,3,3").split(","))
systems = {"gs": bench.build_workload(wl, reps), }
import copy
sj = copy.copy(systems["gs"]); sj.settings = copy.copy(sj.settings); sj.settings.polar_gs_ranked = 0
systems["jac"] = sj
variants = [v.split(":") for v in os.environ.get("LAB_VARIANTS", "jac:0,jac:1,jac:2,jac:4,jac:8,jac:6,jac:14,jac:16,gs:0,gs:1").split(",")]
for mode, abl in variants:
    os.environ["POLAR_ABLATE"] = abl
    p = pkg.pair_from_system(systems[mode])
    for _ in range(2): out = p.compute_resident()
    t = []
    for _ in range(5):
        out = p.compute_resident(); t.append(out["ms_solve"])
    print(f"{mode} ablate={abl:>3s} solve {sum(t)/len(t):7.3f} ms  per-sweep {1e3*sum(t)/len(t)/out['sweeps']:7.1f} us  colors {out['ncolors']} list {out['ms_list']:.2f} lj {out['ms_ljcoul']:.2f}", flush=True)
    p.close()
