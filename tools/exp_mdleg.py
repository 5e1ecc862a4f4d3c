#!/usr/bin/env python3
"""Lab: device-time breakdown of an MD-shaped step (positions re-uploaded, outputs downloaded) beside the resident step."""
import importlib, os, sys, time
import ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module(bench.PKG); wl = importlib.import_module(bench.PKG + ".workload")
s = bench.build_workload(wl, (5, 5, 4), [], solver=bench.CONFIGS[2]["solver"])
p = pkg.pair_from_system(s)
keys = ("ms_total", "ms_list", "ms_ljcoul", "ms_static", "ms_solve", "ms_force", "sweeps")
for _ in range(3): out = p.compute_resident()
print("resident      ", {k: round(out[k], 3) for k in keys}, flush=True)
n, nall = s.nlocal, s.nlocal + s.nghost
f = np.zeros((nall, 3)); mu = np.zeros((n, 3)); ef = np.zeros((n, 3))
dp = C.POINTER(C.c_double); res = pkg.Result()
rng = np.random.default_rng(7); disp = np.zeros_like(s.x)
for mode in ("same positions", "moved 0.01", "moved 0.01", "moved 0.01"):
    if mode != "same positions":
        disp[:n] += rng.normal(scale=0.01, size=(n, 3)); disp[n:] = disp[s.owner[n:]]
    x = np.ascontiguousarray(s.x + disp)
    t0 = time.perf_counter()
    p.set_box(s.boxlo, s.prd); p.set_atoms(s.nlocal, s.nghost, x, s.q, s.alpha, s.type, s.molecule)
    t1 = time.perf_counter()
    p._ck(p.L.polar_compute(p.h, 1, 2, f.ctypes.data_as(dp), mu.ctypes.data_as(dp), ef.ctypes.data_as(dp), C.byref(res)))
    t2 = time.perf_counter()
    out = pkg._result_dict(res)
    print(f"{mode:14s}", {k: round(out[k], 3) for k in keys}, f"set_atoms {1e3*(t1-t0):.2f} ms compute call {1e3*(t2-t1):.2f} ms", flush=True)
out = p.compute_resident()
print("resident again", {k: round(out[k], 3) for k in keys})
