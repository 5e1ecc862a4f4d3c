#!/usr/bin/env python3
"""round 5 (priced before building): how much of k_ljcoul's time is the LOCALITY of its {x,y,z,q} gathers?
The bench box is in LAMMPS `replicate` order (replica by replica, file order inside a replica): a row's ~1,000 partners lie in
up to 8 replicas, i.e. scattered over ~350 KB of the 32-byte record table.  Here the SAME box is handed over with its atoms in
cell order (6.4 A cells, what `atom_modify sort` gives a real LAMMPS run every 1000 steps): the library is unchanged, the list
rows come out sorted by partner index = by cell.  k_ljcoul alone (POLAR_NO_OVERLAP=1) and the whole step, both orders."""
import importlib, json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module(bench.PKG); wl = importlib.import_module(bench.PKG + ".workload")
reps = tuple(int(v) for v in os.environ.get("REPS", "5x5x4").split("x"))
args = ["use_previous", "no", "polar_gs_ranked", "yes", "dd_cutoff", repr(bench.CUT_COUL)] + bench.PREC11


def build(order_mode):
    z = np.load(os.path.join(ROOT, "tests", "golden", "mof5_h2.npz"))
    meta = json.loads(str(z["meta"]))
    st = wl.parse_pair_style_args(list(meta["pair_style_args"]) + args)
    x0, prd0 = z["x"], z["prd"]
    molmax = int(z["molecule"].max())
    xs, mols, r = [], [], 0
    for iz in range(reps[2]):
        for iy in range(reps[1]):
            for ix in range(reps[0]):
                xs.append(x0 + np.array([ix, iy, iz]) * prd0); mols.append(z["molecule"] + r * molmax); r += 1
    nrep = int(np.prod(reps))
    x = np.concatenate(xs); mol = np.concatenate(mols)
    q, alpha, typ = np.tile(z["q"], nrep), np.tile(z["alpha"], nrep), np.tile(z["type"], nrep)
    prd = prd0 * np.array(reps)
    if order_mode == "cells":
        w = 6.4
        nc = np.maximum(1, np.floor(prd / w)).astype(np.int64)
        c = np.minimum(nc - 1, np.floor((x - z["boxlo"]) / prd * nc).astype(np.int64))
        key = (c[:, 2] * nc[1] + c[:, 1]) * nc[0] + c[:, 0]
        o = np.argsort(key, kind="stable")
        x, mol, q, alpha, typ = x[o], mol[o], q[o], alpha[o], typ[o]
    coeff_rows = [[str(int(c[0])), str(int(c[1])), repr(float(c[2])), repr(float(c[3])), repr(float(c[4]))] for c in z["pair_coeff"]]
    g = wl.ewald_g(1.0e-4, q, st.cut_coul, prd)
    return wl.make_system(x, q, alpha, typ, mol, z["boxlo"], prd, meta["ntypes"], coeff_rows, st, g, bonds=None, exclude_intra=True, skin=2.0, name=order_mode)


res = {}
for mode in ("replica", "cells"):
    t0 = time.time()
    s = build(mode)
    print(f"{mode}: system built in {time.time() - t0:.0f} s, {s.nlocal} atoms, {len(s.neigh)} half-list entries", flush=True)
    row = {}
    for overlap in (True, False):
        if not overlap:
            os.environ["POLAR_NO_OVERLAP"] = "1"
        p = pkg.pair_from_system(s)
        out, dt, ms_solve, launches = bench.timed_steps(torch, p, 10, 2)
        p.close()
        os.environ.pop("POLAR_NO_OVERLAP", None)
        row["overlapped" if overlap else "alone"] = {"ms_per_step": 1e3 * dt / 10, "ms_ljcoul": out["ms_ljcoul"], "ms_solve": out["ms_solve"], "sweeps": out["sweeps"],
                                                      "us_per_sweep_launch": 1e3 * ms_solve / max(launches, 1), "eng_pol": out["eng_pol"], "eng_coul": out["eng_coul"], "eng_vdwl": out["eng_vdwl"]}
    res[mode] = row
    print(mode, json.dumps(row), flush=True)
print(json.dumps(res))
