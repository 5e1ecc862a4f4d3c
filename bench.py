#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native polarization pair style.

metric   : atom-steps/s (BASELINE.json), one "step" = one full Pair::compute pass
           (list build, rank metric, LJ+coul, static field, dipole solve, forces, virial)
workload : BASELINE.json configs[1]: "32k-atom replicated polarizable box" = the MOF5+H2 example
           cell replicated 3x3x3 (36,423 atoms, LAMMPS `replicate` semantics), exponential damping,
           fixed_iteration yes max_iterations 30 (31 sweeps), ranked GS, cutoff mode
           r_dd = cut_coul = 12.8345 A.  Inputs resident in HBM before timing.
N > 1    : weak scaling, 36,423 atoms per GPU (3x3x3N replica), one rank per GPU, dipoles
           all-gathered over RCCL once per sweep (see parallel.py).

Prints ONE JSON line on rank 0 (contract in the task statement) including
  roofline     : dipole-field sweep kernel, algorithmic bytes / launch over measured launch time
  cpu_baseline : the CPU oracle ("port") timed on this host on a bounded sample.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "lammps-induced-dipole-polarization-pair-style_amd"

CUT_COUL = 12.8345
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def build_workload(wl, reps, extra=()):
    """BASELINE configs[1]: the MOF5+H2 example cell (1349 atoms) replicated reps=(nx,ny,nz):
    3x3x3 = 36,423 atoms ("32k-atom replicated polarizable box")."""
    args = ["use_previous", "no", "fixed_iteration", "yes", "max_iterations", "30", "polar_gs_ranked", "yes",
            "dd_cutoff", repr(CUT_COUL)] + list(extra)
    return wl.replicate_fixture(os.path.join(ROOT, "tests", "golden", "mof5_h2.npz"), *reps, extra_args=args)


def cpu_baseline(wl, reps=(3, 2, 2)):
    """Oracle (CPU restatement, 1 core) on a bounded sample of the same workload: the same cell,
    cutoffs and solver settings at a smaller replication (cost per atom is N-independent in
    cutoff mode)."""
    from oracle import oracle

    s = build_workload(wl, reps)
    natoms_sample = s.nlocal
    t = time.time()
    out = oracle.compute(s, eflag=1, vflag=2)
    dt = time.time() - t
    return dict(value=natoms_sample / dt, unit="atom-steps/s", cores=1, kind="port",
                sample=f"1 full step of the {reps[0]}x{reps[1]}x{reps[2]} replica ({natoms_sample} atoms, same cell/settings as the GPU workload), "
                       f"{dt:.1f} s, {out['sweeps']} sweeps; host has {os.cpu_count()} cores"), out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reps", type=int, nargs=3, default=[3, 3, 3], help="replication of the 1349-atom cell per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--synth", type=int, default=0, help="use the SURVEY 8(d) synthetic generator with this many atoms instead of the replicated cell")
    ap.add_argument("--extra", nargs="*", default=[], help="extra pair_style keywords (experiments)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch

    pkg = importlib.import_module(PKG)
    wl = importlib.import_module(PKG + ".workload")
    if pkg.device_count() < 1:
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")

    if world > 1 or os.environ.get("POLAR_FORCE_DIST"):  # the env switch lets a 1-GPU box rehearse the distributed path
        par = importlib.import_module(PKG + ".parallel")
        return par.bench_distributed(args, rank, world, local_rank)

    torch.cuda.set_device(0)
    if args.synth:
        s = wl.synth_system(args.synth, seed=1, extra_args=["fixed_iteration", "yes", "max_iterations", "30", "dd_cutoff", repr(CUT_COUL)] + list(args.extra))
    else:
        s = build_workload(wl, tuple(args.reps), args.extra)
    p = pkg.pair_from_system(s, device=0)

    for _ in range(args.warmup):
        out = p.compute_resident(eflag=1, vflag=2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ms_solve = 0.0
    for _ in range(args.steps):
        out = p.compute_resident(eflag=1, vflag=2)
        ms_solve += out["ms_solve"]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0

    n = s.nlocal
    ms_step = 1e3 * dt / args.steps
    value = n * args.steps / dt
    # dominant kernel: k_field_quad (dipole-field sweep).  One sweep = ncolors launches.
    launches = out["sweeps"] * max(out["ncolors"], 1)
    rows = int(np.count_nonzero(s.alpha[:n]))
    # ALGORITHMIC bytes of one sweep, SURVEY.md 8(d): N*(4*K + 8) + N*104 -- int32 neighbor index per
    # pair, int64 row offset, and the per-row streams (x 24, mu 24, E 24, alpha 8, mu_new 24).  Gathers
    # of x_j / mu_j are not counted.  K*N = dd_pairs (the directed polarizable pairs the launch walks).
    bytes_sweep = 4.0 * out["dd_pairs"] + rows * 112.0
    bytes_launch = bytes_sweep / max(out["ncolors"], 1)
    ms_launch = (ms_solve / args.steps) / launches      # HIP events on the library's stream around the solve
    achieved = bytes_launch / (ms_launch * 1e-3) / 1e9
    # What the kernel actually streams by design: + the cached r^2 of every pair (8 B), a deliberate
    # bytes-for-flops trade: 12 B/pair.
    # (the library drops the r^2 stream -- 4 B/pair, r^2 rebuilt in the kernel -- once 12 B/pair no longer fit
    # the 256 MB Infinity Cache: same rule as build_lists in polar_api.hip)
    stream_b = 12.0 if 12.0 * out["dd_pairs"] < 200.0e6 else 4.0
    stream_launch = (stream_b * out["dd_pairs"] + rows * 112.0) / max(out["ncolors"], 1)
    # HBM bytes per launch from PMC counters cannot be read inside this process; the value below was
    # collected with tools/pmc_traffic.sh on this exact workload (separate --pmc passes, per launch:
    # FETCH_SIZE 25,169 KB -> x2 on gfx950 (MI355X_MICROARCH.md, HBM section), WRITE_SIZE 620 KB;
    # profiles/r01_v32_kfield_quad_traffic_pmc.txt, 4 colour launches per sweep) and is reported only for that
    # workload.  It exceeds the streamed bytes by the atom records every launch re-reads into the L2 of each XCD.
    traffic = 51.0e6 if (tuple(args.reps) == (3, 3, 3) and not args.extra and not args.synth) else None
    line = {
        "metric": "atom-steps/sec", "value": value, "unit": "atom-steps/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": (f"SURVEY 8(d) synthetic generator synth({n}, seed 1), exponential damping, fixed_iteration 30, ranked GS, dd_cutoff=cut_coul={CUT_COUL}"
                                if args.synth else
                                f"BASELINE configs[1]: MOF5+H2 cell replicated {args.reps[0]}x{args.reps[1]}x{args.reps[2]} = {n} atoms, exponential damping, "
                                f"fixed_iteration 30 (31 sweeps), ranked GS, dd_cutoff=cut_coul={CUT_COUL}"),
                   "natoms": n, "sweeps": out["sweeps"], "colors": out["ncolors"], "dd_pairs": out["dd_pairs"],
                   "ms_per_dipole_iteration": (ms_solve / args.steps) / max(out["sweeps"], 1),
                   "ms_solve": ms_solve / args.steps, "ms_ljcoul": out["ms_ljcoul"], "ms_force": out["ms_force"],
                   "ms_static": out["ms_static"], "ms_list": out["ms_list"], "ms_rank": out["ms_rank"],
                   "rms_dmu_last_sweep": out["rms_dmu"], "eng_pol": out["eng_pol"]},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel": "k_field_quad (dipole-field sweep, one launch per colour phase)",
                     "bytes_per_launch": bytes_launch, "ms_per_launch": ms_launch,
                     "streamed_bytes_per_launch": stream_launch,
                     "streamed_frac": stream_launch / (ms_launch * 1e-3) / 1e9 / HBM_PEAK_GBS},
    }
    if not args.no_cpu_baseline:
        line["cpu_baseline"], _ = cpu_baseline(wl)
    print(json.dumps(line))


if __name__ == "__main__":
    main()
