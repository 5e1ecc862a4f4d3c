"""Shared parity metrics (SURVEY.md section 8(d) "Parity gate")."""
import glob
import json
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def force_rel_err(f, fref):
    """max_i |f_i - fref_i| / max(|fref_i|, 1e-3 * median|fref|)"""
    n = np.linalg.norm(fref, axis=1)
    floor = 1e-3 * max(np.median(n), 1e-300)
    return float(np.max(np.linalg.norm(f - fref, axis=1) / np.maximum(n, floor)))


def rel(a, b, floor=1e-12):
    return abs(a - b) / max(abs(b), floor)


def golden_refs(case=None):
    out = []
    for p in sorted(glob.glob(os.path.join(GOLD, "ref_*.npz"))):
        z = np.load(p)
        info = json.loads(str(z["info"]))
        if case is None or info["case"] == case:
            out.append((p, info))
    return out


def load_ref_system(wl, info):
    return wl.load_fixture(os.path.join(GOLD, info["case"] + ".npz"), extra_args=info["extra_args"],
                           ncoultablebits=info["ncoultablebits"], newton=bool(info.get("newton", 1)))
