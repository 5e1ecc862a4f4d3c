#!/usr/bin/env python3
"""Lab: sweep counts of the multi-GPU iteration (colour-phase GS inside a rank, block-Jacobi across ranks) emulated in ONE
process on one GPU: `world` compact handles [own | halo | ghosts] stepped in lock-step with a loopback exchange, exactly
the protocol of parallel.bench_distributed.  Used to try exchange variants before they go into parallel.py.

  LAB_REPS  "2x2x8"     replication of the MOF5+H2 cell (z slabs: one rank per z range)
  LAB_WORLD "8"
  LAB_GLUE  "-1"        >= 0: geometric slabs with clusters below this distance kept on one rank (workload.slab_order)
  LAB_W     "0,0.3,0.5,adaptive"   extrapolation weights for the received halo dipoles; "parts2" / "parts4": exchange
            inside the sweep, after every half / every colour phase (polar_step_sweep_part);
            "phase0" / "phase1" / "phase2" (round 4): the single handle's colouring handed to every shard (polar_set_colors: ONE
            colouring across the cuts), an exchange after every colour phase, delivered 0 / 1 / 2 phases LATE -- the worst case of
            polar_dist_step's lag (there a late exchange may also arrive early)
"""
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
par = importlib.import_module(bench.PKG + ".parallel")

reps = tuple(int(v) for v in os.environ.get("LAB_REPS", "2x2x8").split("x"))
world = int(os.environ.get("LAB_WORLD", "8"))
ws = os.environ.get("LAB_W", "0,0.3,0.5,adaptive").split(",")
PREC = ["fixed_iteration", "no", "precision", "1e-11", "max_iterations", "200"] + os.environ.get("LAB_EXTRA", "").split()   # e.g. LAB_EXTRA="polar_sor 1.15"

sg = bench.build_workload(wl, reps, [], build_list=False, solver=PREC)
n_total = sg.nlocal
p = pkg.pair_from_system(sg, device_neigh=True)
ref = p.compute_resident()
mu_ref = p.download("mu", 3 * n_total).reshape(-1, 3)
ncol_ref, col_ref = p.colors(n_total)
p.close()
print(f"single handle: {n_total} atoms, sweeps {ref['sweeps']}, E_pol {ref['eng_pol']:.9f}", flush=True)

glue_dist = float(os.environ.get("LAB_GLUE", "-1"))   # >= 0: geometric slabs (workload.slab_order) with this cluster distance
grid = [int(v) for v in os.environ.get("LAB_GRID", "").split("x") if v]   # e.g. "2x2x2": bricks instead of z slabs
if len(grid) == 3:
    order, offs = wl.brick_order(sg, grid, glue_dist=max(glue_dist, 0.0))
    sg = wl.permute_locals(sg, order)
    mu_ref = mu_ref[order]; col_ref = col_ref[order]
    counts = [int(offs[r + 1] - offs[r]) for r in range(world)]
elif glue_dist >= 0.0:
    order, key, glue = wl.slab_order(sg, axis=2, glue_dist=glue_dist)
    sg = wl.permute_locals(sg, order)
    mu_ref = mu_ref[order]; col_ref = col_ref[order]
    counts, offs = wl.split_sorted(key[order], world, glue)
else:
    counts, offs = par.split_rows(n_total, world)
reach = float(sg.extra["cutneigh"]) + 1e-6
plan = par.P2PHaloPlan(sg.x[:sg.nlocal], sg.prd, offs, reach)
bes, bufs, shard_ids = [], [], []
for r in range(world):
    lo, hi = int(offs[r]), int(offs[r + 1])
    sc = wl.compact_shard_geometric(sg, np.arange(lo, hi), plan.halo_of(r), reach)
    shard_ids.append(np.concatenate([np.arange(lo, hi), np.asarray(plan.halo_of(r), dtype=np.int64)]))
    pr = pkg.pair_from_system(sc, device_neigh=True, row_range=(0, hi - lo))
    be = par.HipShardBackend(pr, 0, hi - lo, 0, global_count=n_total)
    bes.append(be)
    bufs.append(par.p2p_buffers(be, plan, r, compact_lo=lo))
print(f"{world} shards: own {counts[0]}, halo {plan.counts[0]}, peers of rank 0: {len(plan.peers(0))}", flush=True)
prev = [torch.zeros_like(b["recv"]) for b in bufs]
raw = [torch.zeros_like(b["recv"]) for b in bufs]


def exchange(w):
    for r, be in enumerate(bes):
        be.gather_idx(bufs[r]["idx_out"], bufs[r]["send"])
    for r in range(world):
        for k, q in enumerate(bufs[r]["peers"]):
            kq = bufs[q]["peers"].index(r)
            a, b = 3 * bufs[r]["seg_in"][k], 3 * bufs[r]["seg_in"][k + 1]
            c, d = 3 * bufs[q]["seg_out"][kq], 3 * bufs[q]["seg_out"][kq + 1]
            raw[r][a:b] = bufs[q]["send"][c:d]
    for r, be in enumerate(bes):
        if w:
            bufs[r]["recv"].copy_(raw[r] + w * (raw[r] - prev[r]))   # predicted value of the sweep now starting
        else:
            bufs[r]["recv"].copy_(raw[r])
        prev[r].copy_(raw[r])
        be.scatter_idx(bufs[r]["idx_in"], bufs[r]["recv"])


def snapshot():
    """what every rank would receive if all halo dipoles travelled now"""
    for r, be in enumerate(bes):
        be.gather_idx(bufs[r]["idx_out"], bufs[r]["send"])
    snap = [torch.zeros_like(b["recv"]) for b in bufs]
    for r in range(world):
        for k, q in enumerate(bufs[r]["peers"]):
            kq = bufs[q]["peers"].index(r)
            a, b = 3 * bufs[r]["seg_in"][k], 3 * bufs[r]["seg_in"][k + 1]
            c, d = 3 * bufs[q]["seg_out"][kq], 3 * bufs[q]["seg_out"][kq + 1]
            snap[r][a:b] = bufs[q]["send"][c:d]
    return snap


def deliver(snap):
    for r, be in enumerate(bes):
        be.scatter_idx(bufs[r]["idx_in"], snap[r])


for wspec in ws * 2 if len(grid) == 3 else ws:   # (a first pass may only have enlarged a row pitch: POLAR_RETRY_STEP)
    # "phaseL" or "phaseLeK": per-phase schedule, delivery L phases late, an exchange only after every K-th phase (and the last)
    phase_lag = int(wspec[5:].split("e")[0]) if wspec.startswith("phase") else -1
    phase_every = int(wspec.split("e")[-1]) if wspec.startswith("phase") and "e" in wspec[5:] else 1
    for r, be in enumerate(bes):
        be.pair.set_colors(col_ref[shard_ids[r]] if phase_lag >= 0 else None)
    for be in bes:
        be.begin(1, 2)
    exchange(0.0)
    changes = []
    sweeps = 0
    sweep_events = []
    nparts = int(wspec[5:]) if wspec.startswith("parts") else 1
    pending, g = [], 0
    ncs = [be.pair.colors(1)[0] for be in bes] if phase_lag >= 0 else []
    for sw in range(bes[0].max_it + 1):
        if phase_lag >= 0:
            for c in range(ncol_ref):
                while pending and pending[0][0] <= g - 1 - phase_lag:
                    deliver(pending.pop(0)[1])
                for r, be in enumerate(bes):
                    if c < ncs[r]:
                        be.pair._ck(be.pair.L.polar_step_sweep_phase(be.pair.h, c, 0))
                if (c + 1) % phase_every == 0 or c == ncol_ref - 1:
                    pending.append((g, snapshot()))
                g += 1
        elif nparts > 1:
            for part in range(nparts):
                for be in bes:
                    be.sweep_part(part, nparts)
                if part < nparts - 1:
                    exchange(0.0)
        else:
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for be in bes:
                be.sweep()
            e1.record()
            sweep_events.append((e0, e1))
        tot = sum(be.local_change().clone() for be in bes)
        for be in bes:
            be.sweep_end(tot)
        changes.append(float(tot.item()))
        sweeps += 1
        if phase_lag >= 0:
            if all(be.state()[0] for be in bes):
                break
            continue
        if nparts > 1:
            w = 0.0
        elif wspec == "adaptive":
            w = min(0.9, (changes[-1] / changes[-2]) ** 0.5) if len(changes) >= 3 else 0.0
        else:
            w = float(wspec) if sw >= 1 else 0.0
        exchange(w)
        if all(be.state()[0] for be in bes):
            break
    while pending:
        deliver(pending.pop(0)[1])
    outs = [be.finish() for be in bes]
    torch.cuda.synchronize()
    mu = np.zeros((n_total, 3))
    for r, be in enumerate(bes):
        lo, hi = int(offs[r]), int(offs[r + 1])
        mu[lo:hi] = be.pair.download("mu", 3 * (hi - lo)).reshape(-1, 3)
    err = np.max(np.abs(mu - mu_ref)) / np.max(np.abs(mu_ref))
    rate = (changes[-1] / changes[-6]) ** 0.1 if len(changes) > 6 else float("nan")
    if sweep_events:   # the shards run one after the other on this GPU: time of all / world = what ONE rank computes per sweep
        us = 1e3 * sum(a.elapsed_time(b) for a, b in sweep_events) / len(sweep_events) / world
        o0 = outs[0]
        print(f"           per rank: {us:6.1f} us of sweep kernels per sweep; rank 0 device ms: list {o0['ms_list']:.2f} ljcoul {o0['ms_ljcoul']:.2f} "
              f"static {o0['ms_static']:.2f} force {o0['ms_force']:.2f}", flush=True)
    print(f"w={wspec:9s} sweeps {sweeps:3d}  status {[o['status'] for o in outs][:2]}  E_pol {sum(o['eng_pol'] for o in outs):.9f}  "
          f"mu vs single {err:.2e}  contraction/sweep {rate:.3f}", flush=True)
