#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- the in-library multi-GPU driver (polar_dist_*) with SEVERAL ranks on ONE GPU.

Each rank is a thread of this process with its own compact handle [own | halo | ghosts] and its own polar_dist driver;
the RCCL entry points come from tests/dist_mock/fake_rccl.cpp (POLAR_RCCL_LIB), where a send / receive pair is a
device-to-device copy and an all-reduce goes through the host.  Shards, halo plans and send / receive lists are built
exactly as bench.py --gpus N builds them (parallel.bench_distributed).  Prints one JSON line; run by
tests/test_gpu_parity.py::test_in_library_driver_* in a process of its own (the stand-in must be the first "RCCL" the
library opens).

usage: run_mock_dist.py <world> <solver: precision|fixed|jacobi> [reduce_every] [schedule]
  schedule  legacy     every rank colours for itself, one exchange per sweep (round 3)
            lag0/lag1  ONE colouring built by the ranks together (turns by class), per-phase exchanges, a phase waiting for
                       the exchange issued 1 / 2 phases earlier
            imposed0/1 the same schedule with the single handle's own colouring handed to the shards (polar_set_colors):
                       with lag 0 the ranks together run the single-GPU iteration
            badinput   rank 1 is given a setting the driver refuses: every rank must come back with an error, none may hang
            accel1     lag1 with `polar_accel 4` on every rank (Anderson mixing with all-reduced dot products)
            accelL     the legacy schedule (one exchange of all halo dipoles per sweep, one stream) with `polar_accel 4`
            md0/md1    five steps with every atom moved between them: own positions uploaded per rank
                       (polar_set_positions_range), halo positions and ghost images through polar_dist_positions"""
import importlib
import json
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
PKG = "lammps-induced-dipole-polarization-pair-style_amd"
pkg = importlib.import_module(PKG)
wl = importlib.import_module(PKG + ".workload")
par = importlib.import_module(PKG + ".parallel")

world = int(sys.argv[1])
solver = sys.argv[2]
reduce_every = int(sys.argv[3]) if len(sys.argv) > 3 else 1
schedule = sys.argv[4] if len(sys.argv) > 4 else "lag1"
md = schedule.startswith("md")
lag = -1 if schedule in ("legacy", "badinput", "accelL") else int(schedule[-1])
imposed = schedule.startswith("imposed")
extra = {"precision": ["polar_gs_ranked", "yes", "fixed_iteration", "no", "precision", "1e-11", "max_iterations", "200"],
         "fixed": ["polar_gs_ranked", "yes", "fixed_iteration", "yes", "max_iterations", "12"],
         "jacobi": ["polar_gs_ranked", "no", "polar_gs", "no", "fixed_iteration", "yes", "max_iterations", "6"]}[solver]
gold = os.path.join(ROOT, "tests", "golden", "mof5_h2.npz")
# (MOCK_REPS / MOCK_DD: other boxes for tools/, e.g. BASELINE configs[4] "7x7x8" with dd_cutoff 12.8345 on 8 ranks)
reps = tuple(int(v) for v in os.environ.get("MOCK_REPS", "2x2x3").split("x"))
args = ["use_previous", "no", "dd_cutoff", os.environ.get("MOCK_DD", "9.0")] + extra
sg = wl.replicate_fixture(gold, *reps, extra_args=args, build_list=False)
n_total = sg.nlocal
nsteps = 5 if md else 2
rng = np.random.default_rng(11)
disps = [np.zeros((n_total, 3))] + [rng.normal(scale=0.02, size=(n_total, 3)) for _ in range(nsteps - 1)]
disps = np.cumsum(disps, axis=0) if md else [np.zeros((n_total, 3))] * nsteps


def moved(s, disp):
    """all atoms of system s (locals by global id ``owner``, ghosts with their owners) displaced by disp[global id]"""
    return np.ascontiguousarray(s.x + disp[np.asarray(s.owner)])


# the unsharded handle
p0 = pkg.pair_from_system(sg, device_neigh=True)
x0 = sg.x.copy()
refs = []
for k in range(nsteps):
    if md and k > 0:
        p0.set_positions(moved(sg, disps[k]))
    ref = p0.compute_resident()
    refs.append(dict(ref, mu=p0.download("mu", 3 * n_total).reshape(-1, 3)))
ncol0, col0 = p0.colors(n_total)
mu_ref = refs[-1]["mu"]
f_ref = p0.download("f", 3 * (sg.nlocal + sg.nghost)).reshape(-1, 3)[:n_total]
p0.close()

# geometric z slabs, sorbate molecules and bonded clusters whole (as bench.py --gpus N)
order, key, glue = wl.slab_order(sg, axis=2, glue_dist=1.6)
sg = wl.permute_locals(sg, order)       # (owner[:n] stays the identity: local atom k has global id k in the NEW order)
mu_ref, f_ref, col0 = mu_ref[order], f_ref[order], col0[order]
disps = [dk[order] for dk in disps]
counts, offs = wl.split_sorted(key[order], world, glue)
reach = float(sg.extra["cutneigh"]) + 1e-6
plan = par.P2PHaloPlan(sg.x[:sg.nlocal], sg.prd, offs, reach)
classes, nclasses = par.rank_classes([plan.peers(r) for r in range(world)])

uid = pkg.PolarDist.unique_id()      # (also loads the stand-in before any thread asks for it)
outs, mus, fs, errs, locs, hist = [None] * world, [None] * world, [None] * world, [None] * world, [None] * world, [None] * world


def rank_main(r):
    try:
        lo, hi = int(offs[r]), int(offs[r + 1])
        halo = plan.halo_of(r)
        sc = wl.compact_shard_geometric(sg, np.arange(lo, hi), halo, reach)
        if schedule.startswith("accel"):
            import dataclasses
            sc.settings = dataclasses.replace(sc.settings, polar_accel=4)
        if schedule == "badinput" and r == 1:
            import dataclasses
            sc.settings = dataclasses.replace(sc.settings, dd_cutoff=0.0)     # exact mode: polar_dist_step refuses it
        p = pkg.pair_from_system(sc, device_neigh=True, row_range=(0, hi - lo))
        p._ck(p.L.polar_set_global_count(p.h, n_total))
        d = pkg.PolarDist(uid, r, world, device=0)
        peers = plan.peers(r)
        send_lists = [(np.asarray(plan.send[r][q]) - lo).astype(np.int32) for q in peers]
        recv_lists, at = [], hi - lo
        for q in peers:
            m = len(plan.send[q][r])
            recv_lists.append(np.arange(at, at + m, dtype=np.int32))
            at += m
        d.set_halo(p, peers, send_lists, recv_lists)
        d.set_cadence(reduce_every, 4)
        if imposed:   # the single handle's colouring: own rows and halo rows alike
            ids = np.concatenate([np.arange(lo, hi), np.asarray(halo, dtype=np.int64)])
            p.set_colors(col0[ids])
        d.set_schedule(lag, classes[r], nclasses if lag >= 0 else 0)
        d.set_ghosts(p, *par.ghost_map(sc))
        out = None
        hist[r] = []
        gid = np.asarray(sc.owner)[:hi - lo]
        for k in range(nsteps):          # (a second step: the retry of an outgrown pitch has happened by then, colours are reused)
            if md and k > 0:
                p.set_positions_range(0, hi - lo, sc.x[:hi - lo] + disps[k][gid])
                d.positions(p)
            if k == nsteps - 1:
                d.profile(True)          # the last step carries the timed events of polar_dist_profile
            out = d.step(p, 1, 2)
            if md:
                mu_k = p.download("mu", 3 * (hi - lo)).reshape(-1, 3)
                hist[r].append((float(out["eng_pol"]), mu_k))
        outs[r] = {k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in out.items()}
        outs[r]["npeers"] = len(peers)
        outs[r]["profile"] = d.profile_get()
        outs[r]["comm_count"] = d.comm_count()
        loc = d.local_result()
        locs[r] = {k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in loc.items()}
        mus[r] = p.download("mu", 3 * (hi - lo)).reshape(-1, 3)
        nall = sc.nlocal + sc.nghost
        f = p.download("f", 3 * nall).reshape(-1, 3)
        own = np.zeros((hi - lo, 3))     # forces on own atoms: this rank's rows deposit on own atoms only (full list)
        own += f[:hi - lo]
        fs[r] = own
        nc, colr = p.colors(sc.nlocal)
        outs[r]["colors_own"] = colr[:hi - lo].tolist() if lag >= 0 else None
        d.close(); p.close()
    except Exception as e:  # noqa: BLE001
        errs[r] = repr(e)


threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
for t in threads:
    t.start()
for t in threads:
    t.join(timeout=300)
hung = [r for r, t in enumerate(threads) if t.is_alive()]
if schedule == "badinput":
    print(json.dumps({"errors": errs, "hung": hung}))
    os._exit(0 if not hung else 3)
if hung or any(errs):
    print(json.dumps({"error": errs, "hung": hung}))
    os._exit(3)
mu = np.concatenate(mus)
res = {
    "world": world, "solver": solver, "reduce_every": reduce_every, "schedule": schedule, "natoms": n_total, "classes": classes,
    "classes_ok": all(classes[r] != classes[q] for r in range(world) for q in plan.peers(r)),   # no two peers colour in the same turn
    "ref": {k: refs[-1][k] for k in ("eng_pol", "eng_vdwl", "eng_coul", "sweeps", "iterations", "dd_pairs", "status", "ncolors")},
    "ranks": [{k: o[k] for k in ("eng_pol", "eng_vdwl", "eng_coul", "sweeps", "iterations", "dd_pairs", "status", "exchanges", "allreduces", "npeers", "ncolors", "comm_count", "profile", "ms_solve")} for o in outs],
    "local_sum": {k: float(sum(l[k] for l in locs)) for k in ("eng_pol", "eng_vdwl", "eng_coul")},
    "local_virial_sum": np.sum([l["virial"] for l in locs], axis=0).tolist(), "virial": outs[0]["virial"],
    "mu_err": float(np.max(np.abs(mu - mu_ref)) / np.max(np.abs(mu_ref))),
    "rows": [int(c) for c in counts],
}
if lag >= 0:   # the shared colouring: no two rows of one colour within the colour distance, whichever ranks own them
    from scipy.spatial import cKDTree
    col = np.concatenate([np.asarray(o["colors_own"]) for o in outs])
    prd = np.asarray(sg.prd, dtype=np.float64)
    xw = np.mod(sg.x[:n_total] - np.asarray(sg.boxlo), prd)
    xw = np.where(xw >= prd, 0.0, xw)
    pairs = cKDTree(xw, boxsize=prd).query_pairs(2.4 - 1e-9, output_type="ndarray")
    same = (col[pairs[:, 0]] >= 0) & (col[pairs[:, 0]] == col[pairs[:, 1]])
    res["color_clashes"] = int(np.count_nonzero(same))
    res["polarizable_uncoloured"] = int(np.count_nonzero((sg.alpha[:n_total] != 0) & (col < 0)))
if md:
    res["md"] = []
    for k in range(nsteps):
        mu_k = np.concatenate([hist[r][k][1] for r in range(world)])
        ref_mu = refs[k]["mu"][order]
        res["md"].append({"eng_pol": hist[0][k][0], "eng_pol_ref": refs[k]["eng_pol"],
                          "mu_err": float(np.max(np.abs(mu_k - ref_mu)) / np.max(np.abs(ref_mu)))})
print(json.dumps(res))
