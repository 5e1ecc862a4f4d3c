"""resident single-GPU path with the device-built LJ list at 135k (comparison for the one-rank distributed path)"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench, torch
pkg = importlib.import_module(bench.PKG); wl = importlib.import_module(bench.PKG + ".workload")
for dn in (False, True):
    s = bench.build_workload(wl, (5, 5, 4), solver=bench.PREC11, build_list=not dn)
    p = pkg.pair_from_system(s, device_neigh=dn)
    out, dt, ms_solve, launches = bench.timed_steps(torch, p, 5, 2)
    print("device_neigh", dn, "ms/step", 1e3 * dt / 5, {k: round(out[k], 3) for k in ("ms_total", "ms_list", "ms_ljcoul", "ms_static", "ms_solve", "ms_force")}, flush=True)
    p.close()
