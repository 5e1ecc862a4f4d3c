#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- the in-library multi-GPU driver (polar_dist_*) with SEVERAL ranks on ONE GPU.

Each rank is a thread of this process with its own compact handle [own | halo | ghosts] and its own polar_dist driver;
the RCCL entry points come from tests/dist_mock/fake_rccl.cpp (POLAR_RCCL_LIB), where a send / receive pair is a
device-to-device copy and an all-reduce goes through the host.  Shards, halo plans and send / receive lists are built
exactly as bench.py --gpus N builds them (parallel.bench_distributed).  Prints one JSON line; run by
tests/test_gpu_parity.py::test_in_library_driver_with_several_ranks_on_a_mock_transport in a process of its own (the
stand-in must be the first "RCCL" the library opens).

usage: run_mock_dist.py <world> <solver: precision|fixed|jacobi> [reduce_every]"""
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
extra = {"precision": ["polar_gs_ranked", "yes", "fixed_iteration", "no", "precision", "1e-11", "max_iterations", "200"],
         "fixed": ["polar_gs_ranked", "yes", "fixed_iteration", "yes", "max_iterations", "12"],
         "jacobi": ["polar_gs_ranked", "no", "polar_gs", "no", "fixed_iteration", "yes", "max_iterations", "6"]}[solver]
gold = os.path.join(ROOT, "tests", "golden", "mof5_h2.npz")
args = ["use_previous", "no", "dd_cutoff", "9.0"] + extra
sg = wl.replicate_fixture(gold, 2, 2, 3, extra_args=args, build_list=False)
n_total = sg.nlocal

# the unsharded handle
p0 = pkg.pair_from_system(sg, device_neigh=True)
ref = p0.compute_resident()
mu_ref = p0.download("mu", 3 * n_total).reshape(-1, 3)
f_ref = p0.download("f", 3 * (sg.nlocal + sg.nghost)).reshape(-1, 3)[:n_total]
p0.close()

# geometric z slabs, sorbate molecules and bonded clusters whole (as bench.py --gpus N)
order, key, glue = wl.slab_order(sg, axis=2, glue_dist=1.6)
sg = wl.permute_locals(sg, order)
mu_ref, f_ref = mu_ref[order], f_ref[order]
counts, offs = wl.split_sorted(key[order], world, glue)
reach = float(sg.extra["cutneigh"]) + 1e-6
plan = par.P2PHaloPlan(sg.x[:sg.nlocal], sg.prd, offs, reach)

uid = pkg.PolarDist.unique_id()      # (also loads the stand-in before any thread asks for it)
outs, mus, fs, errs = [None] * world, [None] * world, [None] * world, [None] * world


def rank_main(r):
    try:
        lo, hi = int(offs[r]), int(offs[r + 1])
        sc = wl.compact_shard_geometric(sg, np.arange(lo, hi), plan.halo_of(r), reach)
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
        d.set_halo(peers, send_lists, recv_lists)
        d.set_cadence(reduce_every, 4)
        out = None
        for _ in range(2):               # (a second step: the retry of an outgrown pitch has happened by then, colours are reused)
            out = d.step(p, 1, 2)
        outs[r] = {k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in out.items()}
        outs[r]["npeers"] = len(peers)
        mus[r] = p.download("mu", 3 * (hi - lo)).reshape(-1, 3)
        nall = sc.nlocal + sc.nghost
        f = p.download("f", 3 * nall).reshape(-1, 3)
        own = np.zeros((hi - lo, 3))     # forces on own atoms: this rank's rows deposit on own atoms only (full list)
        own += f[:hi - lo]
        fs[r] = own
        d.close(); p.close()
    except Exception as e:  # noqa: BLE001
        errs[r] = repr(e)


threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
for t in threads:
    t.start()
for t in threads:
    t.join(timeout=240)
hung = [r for r, t in enumerate(threads) if t.is_alive()]
if hung or any(errs):
    print(json.dumps({"error": errs, "hung": hung}))
    os._exit(3)
mu = np.concatenate(mus)
res = {
    "world": world, "solver": solver, "reduce_every": reduce_every, "natoms": n_total,
    "ref": {k: ref[k] for k in ("eng_pol", "eng_vdwl", "eng_coul", "sweeps", "iterations", "dd_pairs", "status")},
    "ranks": [{k: o[k] for k in ("eng_pol", "eng_vdwl", "eng_coul", "sweeps", "iterations", "dd_pairs", "status", "exchanges", "allreduces", "npeers")} for o in outs],
    "mu_err": float(np.max(np.abs(mu - mu_ref)) / np.max(np.abs(mu_ref))),
    "rows": [int(c) for c in counts],
}
print(json.dumps(res))
