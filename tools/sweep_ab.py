#!/usr/bin/env python3
"""Lab: A/B of the list-mode sweep kernels on the bench workloads, one process, same box.

  LAB_CASES   comma list of  <reps>:<mode>   e.g. "3x3x3:fixed,5x5x4:prec"
  LAB_KERNELS comma list of  name=ENV1=v;ENV2=v  (env applied before the handle is created)

For every case the first kernel is the reference: forces / dipoles of the others are compared to it.
Scratch tool: results to stdout."""
import copy
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

pkg = importlib.import_module(bench.PKG)
wl = importlib.import_module(bench.PKG + ".workload")

cases = os.environ.get("LAB_CASES", "3x3x3:fixed,5x5x4:prec").split(",")
kernels = os.environ.get("LAB_KERNELS", "quad=POLAR_SWEEP_KERNEL=0,lp=POLAR_SWEEP_KERNEL=2").split(",")
steps = int(os.environ.get("LAB_STEPS", "5"))
print(f"LAB_CASES={','.join(cases)}  LAB_KERNELS={','.join(kernels)}  LAB_STEPS={steps}  library {pkg.kernel_version()}", flush=True)
touched = set()
for case in cases:
    reps, mode = case.split(":")
    reps = tuple(int(v) for v in reps.split("x"))
    extra = [] if mode == "fixed" else ["fixed_iteration", "no", "precision", "1e-11", "max_iterations", "100"]
    s = bench.build_workload(wl, reps, extra)
    ref = None
    for kspec in kernels:
        name, _, envs = kspec.partition("=")
        for k in touched:
            os.environ.pop(k, None)
        import dataclasses
        sk = s
        for e in filter(None, envs.split(";")):
            if e.startswith("kw:"):   # a pair_style keyword of the settings object, e.g. kw:polar_sor:1.15 / kw:deterministic:1
                _, key, val = e.split(":")
                sk = copy.copy(sk)
                sk.settings = dataclasses.replace(sk.settings, **{key: type(getattr(sk.settings, key))(float(val))})
                continue
            k, _, v = e.partition("=")
            os.environ[k] = v
            touched.add(k)
        p = pkg.pair_from_system(sk, lab=os.environ.get("LAB_LIB", "1") != "0")   # the lab build has the kernels and knobs compared here
        for _ in range(2):
            out = p.compute_resident()
        t, tt = [], []
        for _ in range(steps):
            out = p.compute_resident()
            t.append(out["ms_solve"]); tt.append(out["ms_total"])
        f = p.download("f", 3 * (s.nlocal + s.nghost)).reshape(-1, 3)[:s.nlocal]
        mu = p.download("mu", 3 * s.nlocal).reshape(-1, 3)
        err = ""
        if ref is None:
            ref = (f, mu, out["eng_pol"])
        else:
            n = np.linalg.norm(ref[0], axis=1)
            ef = np.max(np.linalg.norm(f - ref[0], axis=1) / np.maximum(n, 1e-3 * np.median(n)))
            em = np.max(np.abs(mu - ref[1])) / np.max(np.abs(ref[1]))
            err = f" | vs {kernels[0].split('=')[0]}: f {ef:.2e} mu {em:.2e} epol {abs(out['eng_pol'] - ref[2]) / abs(ref[2]):.2e}"
        ms = sum(t) / len(t)
        print(f"{case:12s} {name:10s} solve {ms:8.3f} ms  sweeps {out['sweeps']:3d}  per-sweep {1e3 * ms / out['sweeps']:7.1f} us  "
              f"colors {out['ncolors']}  step {sum(tt) / len(tt):7.3f} ms  rms {out['rms_dmu']:.2e} pairs {out['dd_pairs']}{err}", flush=True)
        p.close()
