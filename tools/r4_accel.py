#!/usr/bin/env python3
"""round 4 lab: `polar_accel m` (Anderson mixing on the sweep map) on the bench boxes: sweeps to precision 1e-11, ms per step, dipoles
against the plain iteration's fixed point.  usage: r4_accel.py [config] [m ...]"""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

pkg = importlib.import_module(bench.PKG)
wl = importlib.import_module(bench.PKG + ".workload")
k = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ms = [int(v) for v in sys.argv[2:]] or [0, 1, 2, 3, 4, 5, 6, 8]
cfg = bench.CONFIGS[k]
mu0 = None
for m in ms:
    extra = ("polar_accel", str(m)) if m else ()
    s = bench.build_workload(wl, cfg["reps"], extra=extra, solver=bench.PREC11)
    p = pkg.pair_from_system(s)
    out, dt, ms_solve, launches = bench.timed_steps(torch, p, 5, 2)
    mu = p.download("mu", 3 * s.nlocal).reshape(-1, 3)
    if mu0 is None:
        mu0 = mu
    print(f"config {k} m={m}: sweeps {out['sweeps']:3d} status {out['status']} ms/step {1e3 * dt / 5:7.3f} ms_solve {ms_solve / 5:7.3f} "
          f"E_pol {out['eng_pol']:.9f} rms_dmu {out['rms_dmu']:.2e} mu vs m=0 {np.max(np.abs(mu - mu0)) / np.max(np.abs(mu0)):.2e}", flush=True)
    p.close()
