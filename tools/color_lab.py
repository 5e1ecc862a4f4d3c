#!/usr/bin/env python3
"""Lab: iterations to converge (precision 1e-11) and time per sweep vs the colouring distance."""
import importlib, os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module(bench.PKG); wl = importlib.import_module(bench.PKG + ".workload")
import torch
torch.cuda.set_device(0)
s = bench.build_workload(wl, (3, 3, 3), ["fixed_iteration", "no", "precision", "1e-11", "max_iterations", "200"])
for dc in os.environ.get("LAB_DC", "0.5,1.0,1.2,1.6,2.0,2.6,3.2").split(","):
    os.environ["POLAR_COLOR_DIST"] = dc
    p = pkg.pair_from_system(s)
    for _ in range(2): out = p.compute_resident()
    print(f"dc={dc:>4s} colors {out['ncolors']:2d} iterations {out['iterations']:3d} sweeps {out['sweeps']:3d} status {out['status']} "
          f"solve {out['ms_solve']:.2f} ms  per-sweep {1e3*out['ms_solve']/max(out['sweeps'],1):.1f} us  E_pol {out['eng_pol']:.10f} rms {out['rms_dmu']:.2e}", flush=True)
    p.close()
sj = copy.copy(s); sj.settings = copy.copy(s.settings); sj.settings.polar_gs_ranked = 0
p = pkg.pair_from_system(sj); out = p.compute_resident()
print(f"jacobi: iterations {out['iterations']} status {out['status']} E_pol {out['eng_pol']:.10f}")
