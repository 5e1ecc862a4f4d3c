#!/usr/bin/env python3
"""round 5: ONE rank's handle of the 8-slab decomposition of BASELINE configs[4] (66,104 own + 79,824 halo atoms) stepped alone, with and
without `polar_accel 4` (fixed 12 sweeps: the halo dipoles stay at their initial guess -- the numbers mean nothing, the kernel times do):
the sweep launches of a rank now that a3 is over before the solve, and what the Anderson mixing costs per sweep at a rank's size."""
import importlib, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module(bench.PKG); wl = importlib.import_module(bench.PKG + ".workload"); par = importlib.import_module(bench.PKG + ".parallel")
reps = tuple(int(v) for v in os.environ.get("LAB_REPS", "7x7x8").split("x"))
world = int(os.environ.get("LAB_WORLD", "8"))
accel = os.environ.get("LAB_ACCEL", "0")
FIX = ["fixed_iteration", "yes", "max_iterations", "11", "deterministic", "no"] + (["polar_accel", accel] if accel != "0" else [])
sg = bench.build_workload(wl, reps, [], build_list=False, solver=FIX)
order, key, glue = wl.slab_order(sg, axis=2, glue_dist=1.6)
sg = wl.permute_locals(sg, order)
counts, offs = wl.split_sorted(key[order], world, glue)
reach = float(sg.extra["cutneigh"]) + 1e-6
plan = par.P2PHaloPlan(sg.x[:sg.nlocal], sg.prd, offs, reach)
r = 0
lo, hi = int(offs[r]), int(offs[r + 1])
sc = wl.compact_shard_geometric(sg, np.arange(lo, hi), plan.halo_of(r), reach)
print(f"rank {r}: own {hi - lo}, halo {len(plan.halo_of(r))}, atoms held {sc.nlocal} + {sc.nghost} ghosts, polar_accel {accel}", flush=True)
p = pkg.pair_from_system(sc, device_neigh=True, row_range=(0, hi - lo))
for k in range(6):
    out = p.compute_resident()
print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in out.items() if k.startswith("ms_") or k in ("sweeps", "dd_pairs")})
