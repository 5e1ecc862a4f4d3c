#!/usr/bin/env python3
"""Lab: rows-per-wave of the streaming sweep kernel (POLAR_ROWS_PER_WAVE) for Jacobi and GS."""
import importlib, os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module(bench.PKG); wl = importlib.import_module(bench.PKG + ".workload")
import torch
torch.cuda.set_device(0)
s = bench.build_workload(wl, (3, 3, 3))
sj = copy.copy(s); sj.settings = copy.copy(s.settings); sj.settings.polar_gs_ranked = 0
for rpw in os.environ.get("LAB_RPW", "-1,0,1,2,4,8").split(","):
    os.environ["POLAR_ROWS_PER_WAVE"] = rpw
    for name, sysm in (("jac", sj), ("gs", s)):
        p = pkg.pair_from_system(sysm)
        for _ in range(2): out = p.compute_resident()
        t = []
        for _ in range(5):
            out = p.compute_resident(); t.append(out["ms_solve"])
        print(f"rpw={rpw:>2s} {name:3s} solve {sum(t)/len(t):6.3f} ms per-sweep {1e3*sum(t)/len(t)/out['sweeps']:6.1f} us  E_pol {out['eng_pol']:.9f} rms {out['rms_dmu']:.2e}", flush=True)
        p.close()
