import copy, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_edges as T
pkg = importlib.import_module("lammps-induced-dipole-polarization-pair-style_amd")
wl = importlib.import_module("lammps-induced-dipole-polarization-pair-style_amd.workload")
from oracle import oracle
oracle.build()
rng = np.random.default_rng(11)
n, L = 90, 16.0
s = T._mini(wl, n=n, seed=11, L=L, cut=7.5)
s2 = copy.copy(s)
s2.tilt, s2.triclinic = (3.1, -2.2, 1.7), 1
fr = rng.uniform(0, 1, (n, 3))
xy, xz, yz = s2.tilt
x = np.stack([fr[:, 0] * L + fr[:, 1] * xy + fr[:, 2] * xz, fr[:, 1] * L + fr[:, 2] * yz, fr[:, 2] * L], axis=1)
s2.x = np.ascontiguousarray(x); s2.nghost = 0
for k in ("q", "alpha", "type", "molecule"):
    setattr(s2, k, np.ascontiguousarray(getattr(s, k)[:n]))
s2.owner = np.arange(n)
s2.ilist = np.zeros(0, np.int32); s2.numneigh = np.zeros(n, np.int32)
s2.firstneigh = np.zeros(n, np.int64); s2.neigh = np.zeros(0, np.int32)
for extra in (["polar_gs_ranked", "no", "fixed_iteration", "yes", "max_iterations", "0"], ["polar_gs_ranked", "no", "fixed_iteration", "yes", "max_iterations", "1"], ["precision", "1e-13", "max_iterations", "200"]):
    s2.settings = wl.parse_pair_style_args(["8.0", "7.5", "damp_type", "exponential", "dd_cutoff", "7.5"] + extra)
    ref = oracle.compute(s2, eflag=1, vflag=2)
    out = pkg.pair_from_system(s2).compute()
    print(extra[:3], "ef", np.max(np.abs(out["ef_static"] - ref["ef_static"])) / np.max(np.abs(ref["ef_static"])),
          "mu", np.max(np.abs(out["mu"] - ref["mu"])) / np.max(np.abs(ref["mu"])),
          "f", np.max(np.abs(out["f"] - ref["f"])) / np.max(np.abs(ref["f"])), "epol", out["eng_pol"], ref["eng_pol"], "pairs", out["dd_pairs"], "it", out["iterations"], ref["iterations"], "status", out["status"], ref["status"], "rms", out["rms_dmu"])
    bad = np.argsort(-np.linalg.norm(out["mu"] - ref["mu"], axis=1))[:4]
    print(" worst mu atoms", bad, np.linalg.norm(out["mu"] - ref["mu"], axis=1)[bad], s2.alpha[bad])
