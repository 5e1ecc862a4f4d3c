#!/usr/bin/env python3
"""round 4: what the in-library driver's schedules cost on ONE rank with real RCCL (no link in it): BASELINE configs[2], a third of
the rows exchanged with the rank itself (RCCL allows a self send / receive inside a group).  plain = polar_compute_resident;
legacy = one exchange of all halo rows per sweep on the compute stream; lag0 / lag1 = one colouring "shared" by the one rank,
boundary rows first, one exchange per colour phase on the communication stream, a phase waiting for the exchange issued 1 / 2
phases earlier.  Same rows, same sweeps: the differences are the exposed cost of the RCCL calls and of the event traffic."""
import importlib, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module(bench.PKG); wl = importlib.import_module(bench.PKG + ".workload")
k = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cfg = bench.CONFIGS[k]
s = bench.build_workload(wl, cfg["reps"], solver=bench.PREC11)
steps = 10


def timed(fn):
    for _ in range(2):
        out = fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        out = fn()
    torch.cuda.synchronize()
    return out, 1e3 * (time.perf_counter() - t0) / steps


p = pkg.pair_from_system(s)
out, ms = timed(lambda: p.compute_resident(1, 2))
print(f"plain              : {ms:7.3f} ms/step  solve {out['ms_solve']:6.3f}  sweeps {out['sweeps']}", flush=True)
p.close()
rows = np.arange(0, s.nlocal, 3, dtype=np.int32)
d = pkg.PolarDist(pkg.PolarDist.unique_id(), 0, 1, device=0)
for name, lag, ncls, red in (("legacy, reduce 1", -1, 0, 1), ("legacy, reduce 2", -1, 0, 2), ("lag0,   reduce 2", 0, 1, 2), ("lag1,   reduce 2", 1, 1, 2), ("lag1,   reduce 1", 1, 1, 1)):
    p = pkg.pair_from_system(s)
    p._ck(p.L.polar_set_list_style(p.h, 0))
    d.set_halo(p, [0], [rows], [rows])
    d.set_cadence(red, 4)
    d.set_schedule(lag, 0, ncls)
    out, ms = timed(lambda: d.step(p, 1, 2))
    print(f"{name:19s}: {ms:7.3f} ms/step  solve {out['ms_solve']:6.3f}  sweeps {out['sweeps']}  exchanges {out['exchanges']}  all-reduces {out['allreduces']}  "
          f"E_pol {out['eng_pol']:.9f}", flush=True)
    p.close()
d.close()
