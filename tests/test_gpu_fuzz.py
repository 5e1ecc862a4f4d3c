"""Seeded random systems through the HIP path against the oracle: box shapes, densities, polarizable patterns, cutoffs and
solver settings the fixtures do not happen to have (rows without a single dipole-dipole partner, atoms without neighbours,
boxes barely two cutoffs wide, very few polarizable atoms, a slab of vacuum).  Sizes the oracle finishes in a second or two."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-7   # relative, forces per atom / dipoles / energies (north_star: 1e-5)

SOLVERS = [
    ["polar_gs_ranked", "yes", "precision", "1e-12", "max_iterations", "200"],
    ["polar_gs_ranked", "no", "polar_gs", "yes", "precision", "1e-12", "max_iterations", "200"],
    ["polar_gs_ranked", "no", "fixed_iteration", "yes", "max_iterations", "5"],        # Jacobi, sweep by sweep
    ["polar_gs_ranked", "yes", "precision", "1e-12", "max_iterations", "200", "damp_type", "none"],
]


def _system(wl, seed):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(300, 2200))
    d = wl.synth(n, seed=seed)
    x = d["x"].copy()
    scale = np.array([1.0, rng.uniform(0.8, 1.3), rng.uniform(0.8, 1.5)])      # non-cubic box, density changed with it
    prd = d["prd"] * scale
    x *= scale
    alpha = d["alpha"].copy()
    mode = seed % 4
    if mode == 0:      # half of the atoms lose their polarizability: rows with short or empty dipole lists
        alpha[rng.random(len(alpha)) < 0.5] = 0.0
    elif mode == 1:    # a handful of polarizable atoms only
        keep = rng.choice(len(alpha), size=7, replace=False)
        m = np.ones(len(alpha), dtype=bool); m[keep] = False
        alpha[m] = 0.0
    elif mode == 2:    # a slab of vacuum: atoms near it have one-sided neighbourhoods, some cells are empty
        sel = (x[:, 2] < 0.35 * prd[2]) | (x[:, 2] > 0.65 * prd[2])
        x, alpha = x[sel], alpha[sel]
        d = dict(d, q=d["q"][sel], type=d["type"][sel], molecule=d["molecule"][sel])
        d["q"] = d["q"] - d["q"].mean()
    cutmax = 0.49 * float(prd.min())
    cut = float(rng.uniform(0.6, 1.0)) * min(cutmax - 2.0, 10.0)     # the box stays wider than two (cutoff + skin)
    listmode = seed % 3 != 0 or len(x) > 900
    extra = ["use_previous", "no", "damp_type", "exponential", "damp", "2.1304"] + SOLVERS[seed % len(SOLVERS)]
    if listmode:
        extra += ["dd_cutoff", repr(float(rng.uniform(0.5, 1.0)) * cut)]
    args = ["2.5", repr(cut)] + extra
    st = wl.parse_pair_style_args(args)
    g = wl.ewald_g(1.0e-4, d["q"], st.cut_coul, prd)
    s = wl.make_system(x, d["q"], alpha, d["type"], d["molecule"], np.zeros(3), prd, d["ntypes"], wl.synth_coeff_rows(), st,
                       g, bonds=None, exclude_intra=True, skin=1.0, name=f"fuzz{seed}")
    return s


@pytest.mark.parametrize("seed", range(32))
def test_random_system_matches_oracle(seed, wl, pkg, oracle):
    from helpers import force_rel_err, rel
    s = _system(wl, seed)
    peratom = seed % 4 == 2                # per-atom tallies on every fourth system
    device_neigh = seed % 5 == 1           # LJ/Coulomb list built on the device on every fifth
    ef, vf = (3, 6) if peratom else (1, 2)
    ref = oracle.compute(s, eflag=ef, vflag=vf)
    out = pkg.pair_from_system(s, device_neigh=device_neigh).compute(eflag=ef, vflag=vf)
    assert out["status"] == ref["status"], (out["status"], ref["status"])
    if ref["status"] != 0:            # both walked into the divergence fallback mu = alpha E (PS.cpp:1227-1235)
        assert out["warning"] != ""
    scale = max(np.max(np.abs(ref["mu"])), 1e-30)
    assert np.max(np.abs(out["mu"] - ref["mu"])) / scale < TOL, seed
    assert np.max(np.abs(out["ef_static"] - ref["ef_static"])) < 1e-9 * max(np.max(np.abs(ref["ef_static"])), 1e-30)
    f = oracle.fold_ghost_forces(out["f"], s.owner, s.nlocal)
    fr = oracle.fold_ghost_forces(ref["f"], s.owner, s.nlocal)
    assert force_rel_err(f, fr) < TOL
    for k in ("eng_vdwl", "eng_coul", "eng_pol"):
        assert rel(out[k], ref[k], 1e-9) < TOL, k
    assert np.max(np.abs(out["virial"] - ref["virial"])) < TOL * max(np.max(np.abs(ref["virial"])), 1e-30)
    if peratom:
        for k in ("eatom", "vatom"):
            a = oracle.fold_ghost_forces(out[k], s.owner, s.nlocal)
            b = oracle.fold_ghost_forces(ref[k], s.owner, s.nlocal)
            assert np.max(np.abs(a - b)) < TOL * max(np.max(np.abs(b)), 1e-30), k


def _check_colouring(s, p, dist=2.4):
    """No two atoms of one colour phase within the colour distance (minimum image), every own polarizable row coloured."""
    nc, col = p.colors(s.nlocal)
    x = np.asarray(s.x[:s.nlocal], dtype=float)
    pol = np.asarray(s.alpha[:s.nlocal]) != 0.0
    assert nc >= 1 and np.all(col[pol] >= 0) and np.all(col[pol] < nc) and np.all(col[~pol] < 0)
    assert set(np.unique(col[pol])) == set(range(nc))
    prd = np.asarray(s.prd, dtype=float)
    worst = np.inf
    for c in range(nc):
        xc = x[col == c]
        for a in range(0, len(xc), 512):
            d = xc[a:a + 512, None, :] - xc[None, :, :]
            d -= prd * np.rint(d / prd)
            r2 = np.einsum("ijk,ijk->ij", d, d)
            r2[np.arange(len(r2)), a + np.arange(len(r2))] = np.inf
            worst = min(worst, float(r2.min()) if r2.size else np.inf)
    assert worst > dist * dist, (np.sqrt(worst), nc)
    return nc


@pytest.mark.parametrize("seed", [1, 4, 5, 7, 8, 11, 13, 16, 20, 23, 29])
def test_device_colouring_is_a_proper_colouring(seed, wl, pkg):
    """The colour phases built on the device (cell-by-cell DSATUR, local repair, Jones-Plassmann fallback): read back through
    polar_get_colors and checked pair by pair on random boxes (empty cells, vacuum slabs, odd cell counts, few rows)."""
    s = _system(wl, seed)
    if not s.settings.dd_cutoff > 0 or not (s.settings.polar_gs or s.settings.polar_gs_ranked):
        pytest.skip("no colour phases: exact mode or Jacobi")
    p = pkg.pair_from_system(s)
    out = p.compute(eflag=1, vflag=2)   # (status 1 = the undamped systems' fallback: the phases were swept all the same)
    nc = _check_colouring(s, p)
    assert nc == out["ncolors"]
    p.close()


def test_device_colouring_on_the_mof_replica_has_four_phases(wl, pkg):
    """MOF5+H2 2 x 2 x 2: the device colouring reaches the four phases of the sequential DSATUR (rounds 1-2 built it on the
    host), and it is a proper colouring at the 2.4 A colour distance."""
    import os
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mof5_h2.npz")
    s = wl.replicate_fixture(gold, 2, 2, 2, extra_args=["use_previous", "no", "dd_cutoff", "12.8345", "precision", "1e-11", "max_iterations", "100"])
    p = pkg.pair_from_system(s)
    out = p.compute(eflag=1, vflag=2)
    assert out["status"] == 0 and out["ncolors"] == 4
    assert _check_colouring(s, p) == 4
    p.close()
