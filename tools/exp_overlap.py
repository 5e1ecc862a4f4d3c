"""step time at 135k with the LJ kernel overlapped (default), serial (POLAR_NO_OVERLAP=1), late fork (POLAR_LJ_LATE=1)"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench, torch
pkg = importlib.import_module(bench.PKG); wl = importlib.import_module(bench.PKG + ".workload")
s = bench.build_workload(wl, (5, 5, 4), solver=bench.PREC11)
for env in ({}, {"POLAR_NO_OVERLAP": "1"}, {"POLAR_LJ_LATE": "1"}):
    for k in ("POLAR_NO_OVERLAP", "POLAR_LJ_LATE"):
        os.environ.pop(k, None)
    os.environ.update(env)
    p = pkg.pair_from_system(s)
    out, dt, ms_solve, launches = bench.timed_steps(torch, p, 8, 2)
    print(env, "ms/step %.3f" % (1e3 * dt / 8), {k: round(out[k], 3) for k in ("ms_total", "ms_list", "ms_ljcoul", "ms_static", "ms_solve", "ms_force")}, flush=True)
    p.close()
