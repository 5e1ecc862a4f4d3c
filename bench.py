#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native polarization pair style.

metric   : atom-steps/s (BASELINE.json), one "step" = one full Pair::compute pass
           (cell sort, list build, LJ + Ewald-real, static field, dipole solve, forces, virial).
N = 1    : headline = BASELINE.json configs[2]: the MOF5+H2 example cell replicated 5x5x4 = 134,900 atoms
           ("131k-atom box"), polar_gs_ranked, precision 1e-11, exponential damping, cutoff mode
           r_dd = cut_coul = 12.8345 A.  Inputs resident in HBM before timing.  The same JSON line carries
             config.config1_36k          configs[1] (3x3x3 = 36,423 atoms, fixed_iteration 30)
             config.config4_529k_one_gpu configs[4]'s box (7x7x8 = 528,808 atoms) on ONE GPU: the denominator of
                                         the ">= 6x at 8 GPUs vs 1" target
             config.config0_1349_exact / exact_10792_replica   exact (reference-semantics) mode, with the rocprofv3 /
                                         FETCH_SIZE figures of its sweep kernel
             config.md_leg*              the headline box driven the way LAMMPS drives it: positions re-uploaded
                                         every step, f / mu downloaded, list re-upload + colour check every 10th step.
N > 1    : strong scaling on a fixed box (BASELINE configs[3], configs[4]): 6x6x6 = 291,384 atoms for N = 2, 4;
           7x7x8 = 528,808 atoms for N = 8; one rank per GPU, spatial slabs, point-to-point halo exchange of the
           dipoles over RCCL (parallel.py / csrc/polar_dist.hip).  --schedule picks how (SCHEDULES below); the headline
           is "legacy" (one exchange per sweep, one stream); the others land in config.schedules; config.calibration =
           per-rank-max device times of the parts of the sweep loop beside DESIGN section 6's prediction.

How it runs (round 5): `python bench.py [--gpus N]` by itself is a LAUNCHER that never touches the GPU: the work runs in
budgeted child processes (N = 1: `--direct`; N > 1: torch.distributed.run), the last complete line is relayed, alternative
schedules are separate jobs.  Under a foreign launcher (WORLD_SIZE set) a rank runs directly and an Emitter guarantees exactly
one line: the measured record exists as soon as the timed steps are over and no later leg can lose it.

Prints ONE JSON line on rank 0 (contract in the task statement) including
  roofline     : dipole-field sweep kernel, algorithmic bytes / launch over the measured launch time
  cpu_baseline : the CPU oracle ("port") timed on this host on a bounded sample (N = 1 only).
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
FP64_PEAK_TFLOPS = 78.6  # MI355X FP64 vector peak; the sweep has no matrix work
FLOP_PER_PAIR = 60.0     # ~45 FP64 flop + exp + rsqrt per directed pair of a sweep
L2_GATHER_TBS = (16.8, 18.8)  # MI355X_MICROARCH.md: chip-wide rate of L2-served 64-byte row gathers (66-73 GB/s per CU x 256)
FIXED30 = ["fixed_iteration", "yes", "max_iterations", "30"]
PREC11 = ["fixed_iteration", "no", "precision", "1e-11", "max_iterations", "100"]
CONFIGS = {  # BASELINE.json configs[k] -> replication of the 1,349-atom MOF5+H2 cell, solver keywords
    1: dict(reps=(3, 3, 3), solver=FIXED30, label="configs[1] 32k-atom replicated box, fixed_iteration 30"),
    2: dict(reps=(5, 5, 4), solver=PREC11, label="configs[2] 131k-atom box, polar_gs_ranked, precision 1e-11"),
    3: dict(reps=(6, 6, 6), solver=PREC11, label="configs[3] 262k-atom box, precision 1e-11"),
    4: dict(reps=(7, 7, 8), solver=PREC11, label="configs[4] 512k-atom box, precision 1e-11"),
}


def build_workload(wl, reps, extra=(), build_list=True, solver=FIXED30):
    """The MOF5+H2 example cell (1349 atoms) replicated reps = (nx, ny, nz) with LAMMPS `replicate` semantics,
    exponential damping, ranked GS, cutoff mode r_dd = cut_coul."""
    args = ["use_previous", "no", "polar_gs_ranked", "yes", "dd_cutoff", repr(CUT_COUL)] + list(solver) + list(extra)
    return wl.replicate_fixture(os.path.join(ROOT, "tests", "golden", "mof5_h2.npz"), *reps, extra_args=args,
                                build_list=build_list)


def describe(cfg, n):
    r = cfg["reps"]
    return (f"BASELINE {cfg['label']}: MOF5+H2 cell replicated {r[0]}x{r[1]}x{r[2]} = {n} atoms, exponential damping, "
            f"ranked GS, dd_cutoff=cut_coul={CUT_COUL}")


def cpu_baseline(wl, solver, reps=(3, 2, 2)):
    """Oracle (CPU restatement, 1 core) on a bounded sample of the same workload: the same cell, cutoffs and solver
    settings at a smaller replication (cost per atom is N-independent in cutoff mode)."""
    from oracle import oracle

    s = build_workload(wl, reps, solver=solver)
    t = time.time()
    out = oracle.compute(s, eflag=1, vflag=2)
    dt = time.time() - t
    return dict(value=s.nlocal / dt, unit="atom-steps/s", cores=1, kind="port",
                sample=f"1 full step of the {reps[0]}x{reps[1]}x{reps[2]} replica ({s.nlocal} atoms, same cell/settings as "
                       f"the GPU workload), {dt:.1f} s, {out['sweeps']} sweeps; host has {os.cpu_count()} cores")


def timed_steps(torch, p, steps, warmup):
    for _ in range(warmup):
        out = p.compute_resident(eflag=1, vflag=2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ms_solve, launches = 0.0, 0
    for _ in range(steps):
        out = p.compute_resident(eflag=1, vflag=2)
        ms_solve += out["ms_solve"]
        launches += out["sweeps"] * max(out["ncolors"], 1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return out, dt, ms_solve, launches


def roofline(s, out, ms_solve, launches, steps, pkg):
    """Dominant kernel: the dipole-field sweep (k_field_lp), one launch per colour phase.
    ALGORITHMIC bytes of one sweep, SURVEY.md 8(d): N*(4*K + 8) + N*104 -- int32 neighbor index per pair, int64 row
    offset, and the per-row streams (x 24, mu 24, E 24, alpha 8, mu_new 24); gathers of x_j / mu_j are not counted.
    K*N = dd_pairs (the directed polarizable pairs a sweep walks).  The kernel streams exactly that index
    (4 B per pair) and nothing else per pair.  Launch time = HIP events recorded on the library's stream around
    the solve / number of sweep launches (it therefore includes launch gaps and the end-of-sweep control launches)."""
    rows = int(np.count_nonzero(s.alpha[:s.nlocal]))
    ncol = max(out["ncolors"], 1)
    bytes_launch = (4.0 * out["dd_pairs"] + rows * 112.0) / ncol
    ms_launch = ms_solve / max(launches, 1)
    achieved = bytes_launch / (ms_launch * 1e-3) / 1e9
    # HBM traffic per launch from PMC counters: read from the committed profile of THIS kernel generation on THIS
    # workload (tools/pmc_traffic.sh writes it); null when the file does not match the built library
    traffic, src, low, valu_busy, issue_busy, waves = None, None, None, None, None, None
    try:
        with open(os.path.join(ROOT, "profiles", "traffic_pmc.json")) as fh:
            t = json.load(fh)
        if t.get("kernel_version") == pkg.kernel_version() and t.get("natoms") == s.nlocal:
            traffic, src, low = float(t["bytes_per_launch"]), t.get("source"), t.get("bytes_per_launch_low")
            valu_busy, issue_busy, waves = t.get("valu_busy"), t.get("issue_busy"), t.get("waves_per_simd")
    except (OSError, ValueError, KeyError):
        pass
    # the other roof SURVEY 8(d) asks for: FP64 vector rate of the pair arithmetic (60 flop-equivalents per directed pair)
    tflops = FLOP_PER_PAIR * out["dd_pairs"] / ncol / (ms_launch * 1e-3) / 1e12
    # what the sweep is closest to: every directed pair is one 64-byte record gathered through L2 (rows padded
    # to whole 64-pair trips: + ~6 %); MI355X_MICROARCH.md gives 16.8-18.8 TB/s chip-wide for L2-served row gathers
    gathered = 64.0 * out["dd_pairs"] / ncol
    l2 = {"bytes_per_launch": gathered, "tb_per_s": gathered / (ms_launch * 1e-3) / 1e12, "guide_tb_per_s": L2_GATHER_TBS,
          "frac": gathered / (ms_launch * 1e-3) / 1e12 / L2_GATHER_TBS[0],
          "what": "64-byte neighbour records gathered per launch (one per directed pair) over the launch time, against the low end of the guide's "
                  "L2-served gather rate: the resource the sweep sits closest to, next to the HBM 'frac' and the FP64 'fp64_frac'"}
    return {"bound": "hbm", "l2_gather": l2, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_low": low, "traffic_source": src, "kernel": f"k_field_lp (dipole-field sweep, one launch per colour phase; {pkg.kernel_version()})",
            "bytes_per_launch": bytes_launch, "ms_per_launch": ms_launch, "launches_per_step": launches / max(steps, 1),
            "fp64_tflops": tflops, "fp64_frac": tflops / FP64_PEAK_TFLOPS,
            "valu_busy": valu_busy, "issue_busy": issue_busy, "waves_per_simd": waves,
            "note": "valu_busy / issue_busy / waves_per_simd: SQ_ACTIVE_INST_VALU, SQ_ACTIVE_INST_ANY, SQ_WAVE_CYCLES over the chip's quad-cycles, from the "
                    "committed PMC passes of this kernel version (null when profiles/traffic_pmc.json is of another build).  The sweep is not bound by "
                    "HBM bytes: it gathers 64-byte records at ~62 GB/s per CU (the L2-served gather rate of a CU) with the FP64 arithmetic "
                    "overlapped underneath, in launches of a quarter sweep whose ramp and drain nothing can fill (DESIGN.md section 4)"}


def sub_config(torch, pkg, wl, k, steps, warmup, device_neigh=False, extra=()):
    cfg = CONFIGS[k]
    s = build_workload(wl, cfg["reps"], extra=extra, solver=cfg["solver"], build_list=not device_neigh)
    p = pkg.pair_from_system(s, device_neigh=device_neigh)
    out, dt, ms_solve, launches = timed_steps(torch, p, steps, warmup)
    rf = roofline(s, out, ms_solve, launches, steps, pkg)
    p.close()
    return {"workload": describe(cfg, s.nlocal) + (", LJ/Coulomb list built on the device" if device_neigh else "") + ("; extra keywords: " + " ".join(extra) if extra else ""),
            "natoms": s.nlocal, "steps": steps, "ms_per_step": 1e3 * dt / steps, "atom_steps_per_s": s.nlocal * steps / dt,
            "sweeps": out["sweeps"], "colors": out["ncolors"], "dd_pairs": out["dd_pairs"], "eng_pol": out["eng_pol"],
            "ms_per_dipole_iteration": ms_solve / steps / max(out["sweeps"], 1), "ms_solve": ms_solve / steps,
            "rms_dmu_last_sweep": out["rms_dmu"], "roofline_frac": rf["frac"], "ms_per_sweep_launch": rf["ms_per_launch"]}


def config0_exact(torch, pkg, wl, steps=10, warmup=2):
    """BASELINE configs[0]: the reference's own example, polarization/examples/MOF5+H2 (1349 atoms), EXACT reference semantics
    (all minimum-image pairs, exact-order ranked Gauss-Seidel, max_iterations 30: the knife edge of SURVEY 8(c)), resident."""
    s, _ = wl.load_fixture(os.path.join(ROOT, "tests", "golden", "mof5_h2.npz"),
                           extra_args=["use_previous", "no", "polar_gs_ranked", "yes", "precision", "1e-11", "max_iterations", "30"])
    p = pkg.pair_from_system(s)
    out, dt, ms_solve, _ = timed_steps(torch, p, steps, warmup)
    p.close()
    return {"workload": "BASELINE configs[0]: polarization/examples/MOF5+H2, 1349 atoms, exact all-pairs mode (reference semantics), ranked GS, "
                        "precision 1e-11, max_iterations 30", "natoms": s.nlocal, "steps": steps, "ms_per_step": 1e3 * dt / steps,
            "atom_steps_per_s": s.nlocal * steps / dt, "iterations": out["iterations"], "status": out["status"],
            "us_per_iteration": 1e3 * ms_solve / steps / max(out["sweeps"], 1), "ms_solve": ms_solve / steps, "eng_pol": out["eng_pol"],
            "reference_cpu_s_per_step": 0.85, "roofline_k_gs_blk": exact_pmc(s.nlocal)}


def exact_iteration_bytes(n, npol, B=256):
    """HBM bytes one iteration of exact mode's block sweep reads at n atoms, npol of them polarizable (n >= 1024: blocks of 256
    atoms): per block launch the block's 256 tensor columns of every POLARIZABLE row (the push skips rows with alpha = 0: their
    field is never read) -- 48 B per (row, column) --, the block's G (lower triangle) and N (3B x 3B doubles), the small vectors."""
    nb = (n + B - 1) // B
    return nb * (48.0 * npol * B + 1.5 * (3 * B) ** 2 * 8.0 + 24.0 * (n + 3 * B))


def exact_pmc(natoms):
    """roofline-shaped fields of k_gs_blk from the committed rocprofv3 passes (profiles/exact_traffic_pmc.json, tools/r5_a.sh +
    tools/r5_exact_summary.py): kernel time from --kernel-trace --stats, traffic from separate FETCH_SIZE / WRITE_SIZE passes."""
    try:
        with open(os.path.join(ROOT, "profiles", "exact_traffic_pmc.json")) as fh:
            t = json.load(fh)
    except (OSError, ValueError):
        return None
    if natoms == t.get("natoms"):
        return {k: t.get(k) for k in ("kernel", "launches_per_iteration", "us_per_launch", "algorithmic_bytes_per_launch", "achieved_gb_s", "peak_gb_s", "frac",
                                      "traffic_bytes_per_launch", "tcc_hit_rate", "share_of_gpu_time", "source")}
    if natoms == 1349 and "config0_1349" in t:
        return dict(t["config0_1349"], kernel=t.get("kernel"), source=t.get("source"))
    return None


def exact_replica(torch, pkg, wl, steps=3, warmup=1):
    """The largest system the reference itself was run on: MOF5+H2 `replicate 2 2 2`, 10,792 atoms,
    EXACT reference semantics (all minimum-image pairs: 116 M ordered pairs, a 5.6 GB packed tensor in HBM), ranked GS to
    1e-11.  The reference binary: 46.5 s per step on one Xeon core (dense matrix 8.4 GB)."""
    s = wl.replicate_fixture(os.path.join(ROOT, "tests", "golden", "mof5_h2.npz"), 2, 2, 2,
                             extra_args=["use_previous", "no", "polar_gs_ranked", "yes", "precision", "1e-11", "max_iterations", "100"])
    p = pkg.pair_from_system(s)
    out, dt, ms_solve, _ = timed_steps(torch, p, steps, warmup)
    p.close()
    npol = int(np.count_nonzero(s.alpha[:s.nlocal]))
    return {"workload": "BASELINE.md section 2: MOF5+H2 replicate 2 2 2, 10,792 atoms, exact all-pairs mode (reference semantics), ranked GS, precision 1e-11",
            "natoms": s.nlocal, "steps": steps, "ms_per_step": 1e3 * dt / steps, "atom_steps_per_s": s.nlocal * steps / dt,
            "iterations": out["iterations"], "status": out["status"], "ms_per_dipole_iteration": ms_solve / steps / max(out["sweeps"], 1),
            "ms_solve": ms_solve / steps, "eng_pol": out["eng_pol"], "reference_cpu_s_per_step": 46.5,
            "speedup_over_reference_cpu": 46.5 / (dt / steps),
            # what an iteration of the exact-order sweep reads (csrc/polar_exact.hpp, k_gs_blk): the tensor rows of the POLARIZABLE atoms
            # once (48 B per (row, column); round 4 counted every row: rows with alpha = 0 are skipped, and the FETCH_SIZE pass of
            # round 5 shows exactly that) and G and N of every 256-atom block, over the time of an iteration INCLUDING the
            # once-per-step build of G and N and the gaps between its 43 dependent launches (HIP events, live)
            "hbm_bytes_per_iteration": exact_iteration_bytes(s.nlocal, npol),
            "hbm_frac": exact_iteration_bytes(s.nlocal, npol) / (1e-3 * ms_solve / steps / max(out["sweeps"], 1)) / 8.0e12,
            "roofline_k_gs_blk": exact_pmc(s.nlocal)}


def synth_config(torch, pkg, wl, natoms, steps, warmup):
    """SURVEY 8(d)'s PRIMARY generator synth(N, seed 1) at configs[2]'s size and settings (a load generator: sorbates are placed
    without regard to the framework, so its sweep count says nothing about the solver; density, list lengths and memory pattern
    are the MOF's).  Lists built on the device."""
    s = wl.synth_system(natoms, seed=1, extra_args=PREC11 + ["dd_cutoff", repr(CUT_COUL)], build_list=False)
    p = pkg.pair_from_system(s, device_neigh=True)
    out, dt, ms_solve, launches = timed_steps(torch, p, steps, warmup)
    rf = roofline(s, out, ms_solve, launches, steps, pkg)
    p.close()
    return {"workload": f"SURVEY 8(d) primary generator synth({s.nlocal}, seed 1), exponential damping, ranked GS, precision 1e-11, dd_cutoff=cut_coul={CUT_COUL}, "
                        "LJ/Coulomb list built on the device", "natoms": s.nlocal, "steps": steps, "ms_per_step": 1e3 * dt / steps,
            "atom_steps_per_s": s.nlocal * steps / dt, "sweeps": out["sweeps"], "colors": out["ncolors"], "dd_pairs": out["dd_pairs"], "status": out["status"],
            "ms_per_dipole_iteration": ms_solve / steps / max(out["sweeps"], 1), "ms_solve": ms_solve / steps, "roofline_frac": rf["frac"],
            "l2_gather_frac": rf["l2_gather"]["frac"], "ms_per_sweep_launch": rf["ms_per_launch"]}


def md_leg(pkg, s, steps=20, every=10, seed=7, device_neigh=False, use_previous=False, motion="jitter", temperature=300.0):
    """The headline box driven the way a LAMMPS run drives the shim (lammps_shim/...:compute): every step polar_set_box and the
    positions go up -- polar_set_positions between two neighbor-list builds, polar_set_atoms (all per-atom arrays) on the
    steps that rebuild them -- and f, mu, E_static come back through polar_compute (host pointers); every `every`-th step the
    neighbor list is handed over again (polar_set_neighbors_csr: re-upload, re-symmetrise) or, with ``device_neigh`` (the
    extension keyword `device_neigh yes`), built by the library itself (polar_build_neighbors); the colour phases are
    re-validated on those steps and rebuilt (on the device) when two atoms of one colour have come too close.
    ``motion``: "jitter" = a thermal-size random displacement per step (inside the skin: the uploaded list stays valid);
    "ballistic" = what moves the colouring: every sorbate molecule (<= 8 atoms) flies with a Maxwell velocity of an H2 at
    ``temperature`` (rigid translation, dt = 1 fs, no forces: molecules do run into the framework, more often than in a real
    run), the framework jitters; only with ``device_neigh`` (the list then follows the positions).
    ``use_previous``: the keyword every example deck of the reference sets (PS.cpp:376-386)."""
    if use_previous:
        import copy
        import dataclasses
        s = copy.copy(s)
        s.settings = dataclasses.replace(s.settings, use_previous=1)
    assert motion == "jitter" or device_neigh
    rng = np.random.default_rng(seed)
    p = pkg.pair_from_system(s, device_neigh=device_neigh)
    x0 = s.x.copy()
    n, nall = s.nlocal, s.nlocal + s.nghost
    disp = np.zeros_like(x0)
    vel = None
    if motion == "ballistic":
        mol = np.asarray(s.molecule[:n])
        ids, inv, counts = np.unique(mol, return_inverse=True, return_counts=True)
        small = counts[inv] <= 8
        # LAMMPS real units: v in A/fs, kinetic energy = m v^2 / 2 * mvv2e (48.88821291^2) kcal/mol
        sigma_v = np.sqrt(0.0019872041 * temperature / (2.016 * 48.88821291 ** 2))
        vmol = rng.normal(scale=sigma_v, size=(len(ids), 3))
        vel = np.where(small[:, None], vmol[inv], 0.0)
    first = p.compute(eflag=1, vflag=2)  # first step of a run: lists, colours, allocations
    ms_first_color = first["ms_color_host"]
    t_plain, t_relist, ms_color, ms_dev, t_up, t_cmp, t_nb, nsw = [], [], [], [], [], [], [], []
    f = np.zeros((nall, 3)); mu = np.ascontiguousarray(first["mu"]); ef = np.zeros((n, 3))   # mu: atom->mu_induced
    import ctypes as C
    dp = C.POINTER(C.c_double)
    res = pkg.Result()
    for k in range(steps):
        if motion == "ballistic":
            disp[:n] += vel                                        # dt = 1 fs
            disp[:n] += np.where(vel[:, :1] == 0.0, 1.0, 0.0) * rng.normal(scale=0.004, size=(n, 3))   # framework: vibration-size walk
        else:
            disp[:n] += rng.normal(scale=0.01, size=(n, 3))
        disp[n:] = disp[s.owner[n:]]  # ghosts are images of their owners
        x = np.ascontiguousarray(x0 + disp)
        f[:] = 0.0
        relist = (k % every) == every - 1
        t0 = time.perf_counter()          # --- what the shim does per step, through the C-ABI ---
        p.set_box(s.boxlo, s.prd)
        if relist:
            p.set_atoms(s.nlocal, s.nghost, x, s.q, s.alpha, s.type, s.molecule)
        else:
            p.set_positions(x)
        t1 = time.perf_counter()
        if relist and device_neigh:
            p.build_neighbors_from_system(s)
        elif relist:
            p.set_neighbors_csr(s.ilist, s.numneigh, s.firstneigh, s.neigh)
        t2 = time.perf_counter()
        p._ck(p.L.polar_compute(p.h, 1, 2, f.ctypes.data_as(dp), mu.ctypes.data_as(dp), ef.ctypes.data_as(dp), C.byref(res)))
        t3 = time.perf_counter()
        (t_relist if relist else t_plain).append(1e3 * (t3 - t0))
        t_up.append(1e3 * (t1 - t0)); t_cmp.append(1e3 * (t3 - t2))
        ms_dev.append(res.ms_total)
        nsw.append(res.sweeps)
        if relist:
            t_nb.append(1e3 * (t2 - t1))
        if res.ms_color_host > 0.0:
            ms_color.append(res.ms_color_host)
    p.close()
    ms_plain = float(np.mean(t_plain))
    ms_rel = float(np.mean(t_relist)) if t_relist else ms_plain
    ms_md = (ms_plain * (every - 1) + ms_rel) / every
    how = "built on the device (device_neigh yes)" if device_neigh else "handed over"
    if use_previous:
        how += ", use_previous yes"
    if motion == "ballistic":
        how += f"; sorbate molecules on ballistic Maxwell trajectories ({temperature:.0f} K, 1 fs steps), framework jitter"
    ms_rebuilds = float(np.sum(ms_color))
    return {"what": f"polar_set_box + polar_set_positions (polar_set_atoms on list steps) + polar_compute(host f, mu, E) per step, neighbor list {how} "
                    f"every {every}th step; wall clock of the C-ABI calls ({nall} atoms incl. ghosts)",
            "steps": steps, "sweeps_per_step": float(np.mean(nsw)), "ms_per_step_md": ms_md, "atom_steps_per_s_md": n / (ms_md * 1e-3), "ms_plain_step": ms_plain,
            "ms_reneighbor_step": ms_rel, "colors_rebuilt": len(ms_color), "ms_per_color_rebuild": (ms_rebuilds / len(ms_color)) if ms_color else 0.0,
            "ms_first_step_colouring": ms_first_color,
            "ms_device_per_step": float(np.mean(ms_dev)), "ms_set_positions_or_atoms": float(np.mean(t_up)),
            "ms_compute_call": float(np.median(t_cmp)), "ms_set_neighbors": float(np.mean(t_nb)) if t_nb else 0.0,
            "color_share_of_run": ms_rebuilds / max(float(np.sum(t_plain) + np.sum(t_relist)), 1e-9)}


def visible_gpus():
    """GPUs a fresh child process sees through the library (the launcher itself never initialises HIP)."""
    import subprocess

    code = (f"import sys, importlib; sys.path.insert(0, {ROOT!r}); "
            f"print('POLAR_GPUS', importlib.import_module({PKG!r}).device_count())")
    try:
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
        for ln in r.stdout.splitlines():
            if ln.startswith("POLAR_GPUS"):
                return int(ln.split()[1])
    except (OSError, subprocess.SubprocessError, ValueError):
        pass
    return 0


# ---- the launcher: never touches the GPU; children are FRESH processes under a wall-clock budget -------------------------------
# Why it exists: the first multi-GPU run on hardware must not be able to lose its numbers.  A child prints its
# headline line (flushed) as soon as the timed steps are over and a superseding line once its extra legs are in; the launcher keeps
# the LAST COMPLETE line, kills a child that outlives its budget (the whole process group: torchrun and its ranks) and still
# prints what it has, marked "extras": "timed out".  Other schedules of the N-rank solve run as SEPARATE child jobs, so a hang in
# one of them costs one sub-object of the record, not the headline.
BUDGET_HEADLINE_S = float(os.environ.get("POLAR_BENCH_BUDGET", "900"))       # one child job: workload build + warmup + steps + extras
BUDGET_SCHEDULE_S = float(os.environ.get("POLAR_BENCH_BUDGET_SCHEDULE", "300"))  # one alternative-schedule job (no extras)
EPOL_TOL = 1e-7   # |E_pol per cell - the one-GPU value| / |.|: beyond this the N-rank line is a wrong result and the exit code says so

# name -> (environment of the ranks, extra pair_style keywords): how `bench.py --gpus N` is asked to exchange the halo dipoles.
# "legacy" is the headline: one stream, one communicator, one exchange of all halo dipoles + one all-reduce per sweep -- the form
# closest to what has run on real RCCL (one rank exchanging with itself).
SCHEDULES = {
    "legacy": ({"POLAR_DIST_LAG": "-1"}, []),
    "lag1": ({"POLAR_DIST_LAG": "1", "POLAR_DIST_SPLIT_COMM": "1"}, []),
    "legacy_accel4": ({"POLAR_DIST_LAG": "-1"}, ["polar_accel", "4"]),
}


_CHILDREN = []   # child jobs of this launcher that are still running (each in a session of its own)
_BEST = [None]   # the best record in hand (dict), for a launcher that is told to stop


def _stop_launcher(signum, frame):
    """SIGTERM / SIGINT to the launcher: its children sit in sessions of their own and would otherwise keep the GPUs -- end them,
    print the record in hand (marked), leave."""
    import signal

    for proc in list(_CHILDREN):
        try:
            os.killpg(proc.pid, signal.SIGKILL)
        except (ProcessLookupError, PermissionError):
            pass
    if _BEST[0] is not None:
        _BEST[0]["extras"] = f"launcher stopped by signal {signum}"
        print(json.dumps(_BEST[0]), flush=True)
    os._exit(128 + signum)


def is_result_line(ln):
    return ln.startswith("{") and '"metric"' in ln and ln.rstrip().endswith("}")


def run_child(cmd, env, budget_s, log=None):
    """Run one child job; returns (last complete result line or None, return code or None when killed, timed_out).
    The child gets its own session so that the whole tree (torchrun + ranks) can be killed; it is never re-executed."""
    import signal
    import subprocess
    import threading

    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=None, text=True, bufsize=1, start_new_session=True)
    _CHILDREN.append(proc)
    last = [None]

    def pump():
        for ln in proc.stdout:
            ln = ln.rstrip("\n")
            if is_result_line(ln):
                try:
                    json.loads(ln)
                    last[0] = ln
                    continue
                except ValueError:
                    pass
            print(ln, file=log or sys.stderr, flush=True)

    t = threading.Thread(target=pump, daemon=True)
    t.start()
    timed_out = False
    try:
        rc = proc.wait(timeout=budget_s)
    except subprocess.TimeoutExpired:
        timed_out, rc = True, None
        for sig in (signal.SIGTERM, signal.SIGKILL):
            try:
                os.killpg(proc.pid, sig)
            except (ProcessLookupError, PermissionError):
                break
            try:
                proc.wait(timeout=10)
                break
            except subprocess.TimeoutExpired:
                continue
    t.join(timeout=5)
    if proc in _CHILDREN:
        _CHILDREN.remove(proc)
    return last[0], rc, timed_out


def schedule_summary(line):
    """What the merged record keeps of one schedule's job."""
    c = line.get("config", {})
    keep = ("sweeps", "iterations", "eng_pol_per_cell", "eng_pol_rel_dev_from_one_gpu", "schedule", "rccl_ranks", "calibration", "sweep_loop")
    out = {"ms_per_step": line.get("ms_per_step"), "value": line.get("value")}
    out.update({k: c.get(k) for k in keep if k in c})
    return out


def rank_command(n, port, argv):
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def free_port():
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch(n, argv, probe=visible_gpus, make_cmd=None, schedules=None, out=None):
    """`python bench.py [--gpus N]` without a launcher around it.  N = 1: ONE child (`--direct`) under a budget.  N > 1: N ranks
    (one per GPU) under torch.distributed.run as a CHILD job for the headline schedule, then one more child job per alternative
    schedule, merged into config.schedules.  Prints exactly ONE JSON line (the last complete line of the headline job, merged)
    and returns the exit code: non-zero when the headline job failed, printed no line, or its E_pol check is off.
    Never prints an N = 1 line for a --gpus N request.  POLAR_DIST_BACKEND=gloo (rehearsal): the ranks share whatever GPUs exist.
    ``probe``, ``make_cmd``: injectable for the CPU tests (fake children)."""
    out = out or sys.stdout
    backend = os.environ.get("POLAR_DIST_BACKEND", "nccl")
    have = probe()
    if have < 1:
        print("bench.py: no MI355X visible: the HIP path has no CPU fallback", file=sys.stderr)
        return 2
    if n > 1 and backend == "nccl" and have < n:
        print(f"bench.py: --gpus {n} needs {n} GPUs, this node shows {have} (RCCL takes one rank per device; "
              "POLAR_DIST_BACKEND=gloo rehearses the ranks on shared devices)", file=sys.stderr)
        return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL between processes needs it on this pool
    env["POLAR_BENCH_LAUNCHER"] = "1"                   # the child may print its headline early and a superseding line later
    dist_mode = n > 1 or bool(os.environ.get("POLAR_FORCE_DIST"))   # (the variable: one RCCL rank rehearses the N-rank path on a 1-GPU box)
    if make_cmd is None:
        make_cmd = (lambda name, extra_argv: rank_command(n, free_port(), list(argv) + list(extra_argv))) if dist_mode \
            else (lambda name, extra_argv: [sys.executable, os.path.abspath(__file__), "--direct"] + list(argv) + list(extra_argv))
    names = list(schedules if schedules is not None else (SCHEDULES if dist_mode else ["single"]))
    if dist_mode and "--schedule" in argv:   # an explicit --schedule: that one job, nothing beside it
        names = [argv[argv.index("--schedule") + 1]] if argv.index("--schedule") + 1 < len(argv) else names
        argv = [a for k, a in enumerate(argv) if a != "--schedule" and (k == 0 or argv[k - 1] != "--schedule")]
    head = names[0]
    t0 = time.time()
    try:
        import signal
        signal.signal(signal.SIGTERM, _stop_launcher)
        signal.signal(signal.SIGINT, _stop_launcher)
    except (ValueError, OSError):   # (not the main thread: tests)
        pass
    # (the ranks themselves turn --schedule NAME into the driver's settings and the extra keywords: SCHEDULES)
    ln, rc, timed_out = run_child(make_cmd(head, ["--schedule", head] if dist_mode else []), env, BUDGET_HEADLINE_S)
    if ln is None:
        why = "timed out" if timed_out else f"exit code {rc}"
        print(f"bench.py: the {'headline' if dist_mode else 'bench'} job ({n} rank(s)) gave no result line ({why})", file=sys.stderr)
        return (rc or 3) if not timed_out else 3
    line = json.loads(ln)
    if timed_out:
        line["extras"] = "timed out"
    elif rc != 0:
        line["extras"] = f"child exit code {rc}"
    exit_code = 0 if (rc == 0 or timed_out) else rc
    _BEST[0] = line
    if dist_mode:
        line.setdefault("config", {})["schedules"] = {head: schedule_summary(line)}
        line["config"]["headline_schedule"] = head
        for name in names[1:]:
            extra = ["--schedule", name, "--no-extras"]
            try:
                sl, src, sto = run_child(make_cmd(name, extra), env, BUDGET_SCHEDULE_S)
                if sl is None:
                    line["config"]["schedules"][name] = {"error": "timed out" if sto else f"no result line (exit code {src})"}
                else:
                    sub = schedule_summary(json.loads(sl))
                    if sto:
                        sub["note"] = "timed out after its line"
                    elif src != 0:
                        sub["exit_code"] = src
                    line["config"]["schedules"][name] = sub
            except Exception as e:  # noqa: BLE001 -- a schedule job must never cost the headline
                line["config"]["schedules"][name] = {"error": repr(e)}
        dev = line["config"].get("eng_pol_rel_dev_from_one_gpu")
        if dev is not None and not (dev <= EPOL_TOL):
            print(f"bench.py: E_pol per cell is {dev:.2e} off the one-GPU value (tolerance {EPOL_TOL:.0e}): wrong result", file=sys.stderr)
            exit_code = exit_code or 4
    line["launcher_wall_s"] = time.time() - t0
    print(json.dumps(line), file=out, flush=True)
    return exit_code


class Emitter:
    """Exactly ONE result line from this process tree, whatever happens after the timed steps.
    headline(line): the measured record exists from here on.  Under this file's own launcher (POLAR_BENCH_LAUNCHER, which relays
    the LAST complete line) it is printed at once and superseded later; under a foreign launcher (the driver's torchrun) it is
    held back.  final(line) prints the complete record.  If the extra legs outlive their budget -- a hang in an RCCL call, a
    stuck leg -- a watchdog thread prints the stored line marked "extras": "timed out in <leg>" and ends the process with exit
    code 0 (every rank arms the same watchdog, so the job ends).  Ranks other than 0 never print."""

    def __init__(self, is_rank0, budget_s):
        import threading

        self.rank0, self.budget = is_rank0, budget_s
        self.early = bool(os.environ.get("POLAR_BENCH_LAUNCHER"))
        self.lock = threading.Lock()
        self.line, self.done, self.stage, self.timer = None, False, "extras", None
        self._threading = threading

    def headline(self, line):
        with self.lock:
            self.line = json.loads(json.dumps(line)) if line is not None else None   # (a private copy: the legs go on editing theirs)
            if self.rank0 and self.early and line is not None:
                print(json.dumps(line), flush=True)
        self.timer = self._threading.Timer(self.budget, self._expire)
        self.timer.daemon = True
        self.timer.start()

    def final(self, line):
        with self.lock:
            if self.done:
                return
            self.done = True
            if self.timer is not None:
                self.timer.cancel()
            if self.rank0 and line is not None:
                print(json.dumps(line), flush=True)

    def _expire(self):
        with self.lock:
            if self.done:
                return
            self.done = True
            if self.rank0 and self.line is not None:
                self.line["extras"] = f"timed out in {self.stage} after {self.budget:.0f} s"
                print(json.dumps(self.line), flush=True)
            sys.stdout.flush()
            os._exit(0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=0, help="BASELINE configs[k] as the headline (default: 2 at N=1; 3 at N=2,4; 4 at N=8)")
    ap.add_argument("--reps", type=int, nargs=3, default=None, help="replication of the 1349-atom cell (experiments; overrides --config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline only: skip the configs[1] / 529k / MD-shaped legs")
    ap.add_argument("--synth", type=int, default=0, help="use the SURVEY 8(d) synthetic generator with this many atoms instead of the replicated cell")
    ap.add_argument("--extra", nargs="*", default=[], help="extra pair_style keywords (experiments)")
    ap.add_argument("--direct", action="store_true", help="N = 1: run in THIS process (profilers, scripts); default: a budgeted child of the launcher")
    ap.add_argument("--schedule", default="", help=f"N > 1: how the ranks exchange the halo dipoles, one of {sorted(SCHEDULES)} (default: legacy)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.schedule and args.schedule not in SCHEDULES:
        raise SystemExit(f"bench.py: --schedule is one of {sorted(SCHEDULES)}")
    if "WORLD_SIZE" not in os.environ and not args.direct:
        # `python bench.py [--gpus N]` by itself: become the launcher.  Nothing here has touched the GPU (no torch, no HIP
        # call): the workers are fresh children under a budget, never a re-exec of an initialised process.
        argv = [a for a in sys.argv[1:]]
        raise SystemExit(launch(args.gpus, argv))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and not os.environ.get("POLAR_FORCE_DIST"):
        # a line that says n_gpus = world for a --gpus N request would be a wrong record: refuse
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} (or run `python bench.py --gpus {args.gpus}`, which starts the ranks itself)")
    import torch

    pkg = importlib.import_module(PKG)
    wl = importlib.import_module(PKG + ".workload")
    if pkg.device_count() < 1:
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")

    if world > 1 or os.environ.get("POLAR_FORCE_DIST"):  # the env switch lets a 1-GPU box rehearse the distributed path
        par = importlib.import_module(PKG + ".parallel")
        return par.bench_distributed(args, rank, world, local_rank)

    torch.cuda.set_device(0)
    k = args.config or 2
    cfg = dict(CONFIGS[k])
    if args.reps:
        cfg["reps"] = tuple(args.reps)
        cfg["label"] = f"(experiment) {cfg['label']}"
    if args.synth:
        s = wl.synth_system(args.synth, seed=1, extra_args=cfg["solver"] + ["dd_cutoff", repr(CUT_COUL)] + list(args.extra))
        workload = f"SURVEY 8(d) synthetic generator synth({s.nlocal}, seed 1), exponential damping, ranked GS, dd_cutoff=cut_coul={CUT_COUL}"
    else:
        s = build_workload(wl, cfg["reps"], args.extra, solver=cfg["solver"])
        workload = describe(cfg, s.nlocal)
    p = pkg.pair_from_system(s, device=0)
    out, dt, ms_solve, launches = timed_steps(torch, p, args.steps, args.warmup)
    p.close()
    n = s.nlocal
    config = {"workload": workload, "natoms": n, "sweeps": out["sweeps"], "iterations": out["iterations"], "colors": out["ncolors"],
              "dd_pairs": out["dd_pairs"], "ms_per_dipole_iteration": (ms_solve / args.steps) / max(out["sweeps"], 1),
              "ms_solve": ms_solve / args.steps, "ms_ljcoul": out["ms_ljcoul"], "ms_force": out["ms_force"],
              "ms_static": out["ms_static"], "ms_list": out["ms_list"], "ms_rank": out["ms_rank"],
              "rms_dmu_last_sweep": out["rms_dmu"], "eng_pol": out["eng_pol"], "kernel_version": pkg.kernel_version()}
    line = {
        "metric": "atom-steps/sec", "value": n * args.steps / dt, "unit": "atom-steps/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic", "config": config,
        "roofline": roofline(s, out, ms_solve, launches, args.steps, pkg),
    }
    if not args.no_cpu_baseline:   # pure host work (the oracle on one core, ~15 s): part of the headline record
        try:
            line["cpu_baseline"] = cpu_baseline(wl, cfg["solver"])
        except Exception as e:  # noqa: BLE001
            line["cpu_baseline"] = {"error": repr(e)}
    # ---- the measured record exists from here on: no later leg can lose it ----
    em = Emitter(True, float(os.environ.get("POLAR_BENCH_EXTRAS_BUDGET", "700")))
    em.headline(line)

    def leg(where, name, fn):
        """One extra leg: an exception (or a watchdog expiry) inside it costs its own sub-object, nothing else."""
        em.stage = name
        t0 = time.time()
        try:
            where[name] = fn()
        except Exception as e:  # noqa: BLE001
            where[name] = {"error": repr(e), "after_s": time.time() - t0}

    def alone():
        # The same sweeps with the chip to themselves: in the timed region a3 (LJ + Ewald-real) runs beside them on its
        # low-priority side stream -- it shortens the step and stretches the sweeps, whose launch time therefore says less
        # about the kernel the faster the other kernels get.  A few extra steps with a3 kept on the main stream
        # (POLAR_NO_OVERLAP): reported as roofline.alone, never as the headline.
        os.environ["POLAR_NO_OVERLAP"] = "1"
        try:
            p2 = pkg.pair_from_system(s, device=0)
            out2, dt2, ms_solve2, launches2 = timed_steps(torch, p2, 5, 2)
            p2.close()
        finally:
            os.environ.pop("POLAR_NO_OVERLAP", None)
        r = roofline(s, out2, ms_solve2, launches2, 5, pkg)
        return {"what": "the same launches with a3 kept off its side stream (5 extra steps): the sweep kernel by itself",
                "ms_per_launch": r["ms_per_launch"], "achieved": r["achieved"], "frac": r["frac"], "ms_per_step": 1e3 * dt2 / 5}

    leg(line["roofline"], "alone", alone)
    if not args.no_extras and not args.synth and not args.reps:
        leg(config, "md_leg", lambda: md_leg(pkg, s))
        leg(config, "md_leg_device_neigh", lambda: md_leg(pkg, s, device_neigh=True))
        leg(config, "md_leg_use_previous", lambda: md_leg(pkg, s, device_neigh=True, use_previous=True))
        leg(config, "md_leg_ballistic", lambda: md_leg(pkg, s, steps=200, device_neigh=True, motion="ballistic", temperature=300.0))
        leg(config, "config0_1349_exact", lambda: config0_exact(torch, pkg, wl))
        leg(config, "exact_10792_replica", lambda: exact_replica(torch, pkg, wl))
        leg(config, "config2_synth_131k", lambda: synth_config(torch, pkg, wl, 131072, steps=min(args.steps, 5), warmup=1))
        leg(config, "config1_36k", lambda: sub_config(torch, pkg, wl, 1, steps=max(args.steps, 10), warmup=args.warmup))
        leg(config, "config1_36k_deterministic_no", lambda: sub_config(torch, pkg, wl, 1, steps=max(args.steps, 10), warmup=args.warmup, extra=("deterministic", "no")))
        leg(config, "config4_529k_one_gpu", lambda: sub_config(torch, pkg, wl, 4, steps=min(args.steps, 5), warmup=1, device_neigh=True))
        # opt-in extension keywords on the headline box (NOT the headline: the reference has neither)
        leg(config, "config2_polar_sor_1p15", lambda: sub_config(torch, pkg, wl, 2, steps=max(args.steps, 10), warmup=args.warmup, extra=("polar_sor", "1.15")))
        leg(config, "config2_polar_accel_4", lambda: sub_config(torch, pkg, wl, 2, steps=max(args.steps, 10), warmup=args.warmup, extra=("polar_accel", "4")))
        leg(config, "config2_deterministic", lambda: sub_config(torch, pkg, wl, 2, steps=max(args.steps, 10), warmup=args.warmup, extra=("deterministic", "yes")))
    em.final(line)


if __name__ == "__main__":
    main()
