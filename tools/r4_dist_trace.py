#!/usr/bin/env python3
"""round 4: ONE rank, real RCCL, the per-phase schedule with lag 0 (a phase waits for the exchange of the phase before it), three
steps -- for rocprofv3 --kernel-trace: does the exchange of a phase's boundary dipoles run beside the phase's interior rows?"""
import importlib, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module(bench.PKG); wl = importlib.import_module(bench.PKG + ".workload")
s = bench.build_workload(wl, bench.CONFIGS[2]["reps"], solver=bench.PREC11)
rows = np.arange(0, s.nlocal, 3, dtype=np.int32)
d = pkg.PolarDist(pkg.PolarDist.unique_id(), 0, 1, device=0)
p = pkg.pair_from_system(s)
p._ck(p.L.polar_set_list_style(p.h, 0))
d.set_halo(p, [0], [rows], [rows])
d.set_cadence(2, 4)
d.set_schedule(int(os.environ.get("LAG", "0")), 0, 1)
for k in range(3):
    out = d.step(p, 1, 2)
torch.cuda.synchronize()
print("sweeps", out["sweeps"], "ms_solve", out["ms_solve"])
