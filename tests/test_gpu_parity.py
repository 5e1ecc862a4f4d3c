"""GPU parity tests proper: the HIP path, called through the C-ABI, against
  (a) golden vectors from the reference's own compiled pair style (tests/golden/ref_*.npz), and
  (b) the CPU oracle on the same inputs.
Bar (BASELINE.json north_star): per-atom forces and energies within 1e-5 relative of the
reference.  The asserts below use 1e-7 -- FP64 kernels should clear the bar by two orders."""
import os

import numpy as np
import pytest

from helpers import GOLD, force_rel_err, golden_refs, load_ref_system, rel

pytestmark = pytest.mark.gpu

TOL = 1e-7        # asserted
BAR = 1e-5        # north_star tolerance (documented bar)
_REFS = golden_refs()


def _run(pkg, s, info):
    p = pkg.pair_from_system(s, modify_args=info.get("modify_args", ()))
    out, mu = None, None
    for _ in range(info.get("ncalls", 1)):
        out = p.compute(eflag=info.get("eflag", 1), vflag=info.get("vflag", 2), mu=mu)
        mu = out["mu"]
    p.close()
    return out


@pytest.mark.parametrize("path,info", _REFS, ids=[f"{i['case']}-{i['variant']}" for _, i in _REFS])
def test_hip_matches_reference_golden(path, info, wl, pkg, oracle):
    z = np.load(path)
    s, _ = load_ref_system(wl, info)
    out = _run(pkg, s, info)
    f = oracle.fold_ghost_forces(out["f"], s.owner, s.nlocal)
    assert force_rel_err(f, z["f"]) < TOL
    scale_mu = max(np.max(np.abs(z["mu"])), 1e-30)
    assert np.max(np.abs(out["mu"] - z["mu"])) / scale_mu < TOL
    assert np.max(np.abs(out["ef_static"] - z["ef_static"])) / np.max(np.abs(z["ef_static"])) < TOL
    e = z["energies"]
    assert rel(out["eng_vdwl"], e[0], 1e-6) < TOL
    assert rel(out["eng_coul"], e[1], 1e-6) < TOL
    assert rel(out["eng_pol"], e[2], 1e-6) < TOL
    assert np.max(np.abs(out["virial"] - z["virial"])) < TOL * max(1.0, np.max(np.abs(z["virial"])))
    assert (out["status"] == 1) == bool(info["warnings"])
    if info["warnings"]:
        assert out["warning"] == info["message"]


def test_iteration_count_matches_reference_knife_edge(wl, pkg):
    """BASELINE config 0: max_iterations 30 converges in exactly 30 sweeps in the reference
    (SURVEY.md 8(c)); one more would trigger the alpha*E fallback."""
    s, meta = wl.load_fixture(os.path.join(GOLD, "mof5_h2.npz"), extra_args=["use_previous", "no", "max_iterations", "30"])
    p = pkg.pair_from_system(s)
    out = p.compute()
    assert out["iterations"] == meta["known"]["iterations"] == 30
    assert out["status"] == 0


@pytest.mark.parametrize("case", ["bulk_h2", "mof5_h2"])
@pytest.mark.parametrize("mode", ["gs", "ranked"])
def test_cutoff_mode_matches_oracle(case, mode, wl, pkg, oracle):
    """dd_cutoff extension (device cell/neighbor lists + colour-phase Gauss-Seidel) against the
    oracle's sequential Gauss-Seidel in the same truncated model."""
    zprd = np.load(os.path.join(GOLD, case + ".npz"))["prd"]
    rdd = 0.4 * float(zprd.min())
    extra = ["use_previous", "no", "dd_cutoff", repr(rdd), "precision", "1e-13", "max_iterations", "200"]
    if mode == "gs":
        extra += ["polar_gs_ranked", "no", "polar_gs", "yes"]
    s, _ = wl.load_fixture(os.path.join(GOLD, case + ".npz"), extra_args=extra)
    ref = oracle.compute(s, eflag=1, vflag=2)
    p = pkg.pair_from_system(s)
    out = p.compute()
    assert out["status"] == 0 and ref["status"] == 0
    f = oracle.fold_ghost_forces(out["f"], s.owner, s.nlocal)
    fr = oracle.fold_ghost_forces(ref["f"], s.owner, s.nlocal)
    assert force_rel_err(f, fr) < TOL
    assert rel(out["eng_pol"], ref["eng_pol"]) < TOL
    assert np.max(np.abs(out["mu"] - ref["mu"])) / np.max(np.abs(ref["mu"])) < TOL
    assert out["ncolors"] > 0


def test_cutoff_mode_jacobi_and_full_cutoff_equals_exact(wl, pkg, oracle):
    """With dd_cutoff >= sqrt(3)/2 L... not reachable with the list path (needs L >= 2 rc); instead
    check list-mode Jacobi against the oracle's Jacobi sweep by sweep (fixed 5 iterations)."""
    extra = ["use_previous", "no", "polar_gs_ranked", "no", "fixed_iteration", "yes", "max_iterations", "5",
             "dd_cutoff", "9.0"]
    s, _ = wl.load_fixture(os.path.join(GOLD, "bulk_h2.npz"), extra_args=extra)
    ref = oracle.compute(s, eflag=1, vflag=2)
    out = pkg.pair_from_system(s).compute()
    assert out["sweeps"] == ref["sweeps"] == 6 and out["iterations"] == ref["iterations"] == 5
    assert np.max(np.abs(out["mu"] - ref["mu"])) / np.max(np.abs(ref["mu"])) < TOL
    f = oracle.fold_ghost_forces(out["f"], s.owner, s.nlocal)
    fr = oracle.fold_ghost_forces(ref["f"], s.owner, s.nlocal)
    assert force_rel_err(f, fr) < TOL


def test_resident_compute_and_row_pointer_lists(wl, pkg, oracle):
    s, _ = wl.load_fixture(os.path.join(GOLD, "bulk_h2.npz"), extra_args=["use_previous", "no"])
    p = pkg.pair_from_system(s)
    a = p.compute()
    # LAMMPS-style int** rows
    rows = [s.neigh[s.firstneigh[i]:s.firstneigh[i] + s.numneigh[i]] for i in range(s.nlocal)]
    p.set_neighbors_rows(s.ilist, s.numneigh, rows)
    b = p.compute_resident()
    fb = p.download("f", 3 * (s.nlocal + s.nghost)).reshape(-1, 3)
    assert rel(b["eng_pol"], a["eng_pol"]) < 1e-12 and rel(b["eng_vdwl"], a["eng_vdwl"]) < 1e-12
    assert force_rel_err(fb[:s.nlocal], a["f"][:s.nlocal]) < 1e-10


def test_no_silent_fallback_library_is_loaded(pkg):
    """The product path is the in-tree HIP library; make sure it is what is mapped."""
    assert pkg.device_count() >= 1
    with open("/proc/self/maps") as fh:
        assert any("libpolar_mi355x.so" in ln for ln in fh)
