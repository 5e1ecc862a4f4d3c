"""BASELINE configs[1] at its full size (36,423 atoms, the workload bench.py times) through
size-independent properties -- the oracle needs minutes for this system, the properties need none:

  * translation symmetry: the box is 3x3x3 copies of one cell, so the 27 images of an atom must
    get the same dipole, static field and force;
  * Newton's third law: every force term on the path is pairwise antisymmetric, so the forces
    (ghost contributions folded back, as reverse_comm does) sum to zero;
  * the reference's own self-check (PS.cpp:395-404 against :632): at the fixed point
    u_self + u_ef + u_dd == -1/2 sum_i E_static,i . mu_i;
  * per-atom tallies add up to the global ones.
"""
import os

import numpy as np
import pytest

from helpers import GOLD

pytestmark = pytest.mark.gpu

CUT = "12.8345"


@pytest.fixture(scope="module")
def full(wl, pkg):
    extra = ["use_previous", "no", "fixed_iteration", "yes", "max_iterations", "30", "polar_gs_ranked", "yes",
             "dd_cutoff", CUT]
    s = wl.replicate_fixture(os.path.join(GOLD, "mof5_h2.npz"), 3, 3, 3, extra_args=extra)
    p = pkg.pair_from_system(s)
    out = p.compute(eflag=3, vflag=5)
    p.close()
    return s, out


def test_images_of_an_atom_agree(full):
    s, out = full
    n0 = s.nlocal // 27
    f = np.zeros((s.nlocal, 3))
    np.add.at(f, s.owner, out["f"])
    for name, a in (("mu", out["mu"]), ("ef_static", out["ef_static"]), ("f", f)):
        img = a.reshape(27, n0, 3)
        scale = np.max(np.linalg.norm(img[0], axis=1))
        dev = np.max(np.linalg.norm(img - img[0][None], axis=2)) / scale
        assert dev < 1e-8, (name, dev)  # FP64 with 31 sweeps: solver residual 2e-11, summation order differs


def test_forces_sum_to_zero(full):
    s, out = full
    tot = out["f"].sum(axis=0)
    assert np.max(np.abs(tot)) < 1e-9 * np.abs(out["f"]).sum()


def test_energy_identity_at_the_fixed_point(full):
    s, out = full
    assert out["sweeps"] == 31 and out["status"] == 0
    lhs = out["u_self"] + out["u_ef"] + out["u_dd"]
    rhs = -0.5 * float(np.sum(out["ef_static"] * out["mu"]))
    assert abs(lhs - out["eng_pol"]) < 1e-12 * abs(lhs)
    assert abs(lhs - rhs) < 1e-8 * abs(rhs)


def test_peratom_tallies_add_up(full):
    s, out = full
    assert abs(out["eatom"].sum() - (out["eng_vdwl"] + out["eng_coul"])) < 1e-9 * abs(out["eng_vdwl"] + out["eng_coul"])
    assert np.max(np.abs(out["vatom"].sum(axis=0) - out["virial"])) < 1e-9 * np.max(np.abs(out["virial"]))
    # 27 identical cells: the per-cell energy is 1/27 of the total
    n0 = s.nlocal // 27
    ea = np.zeros(s.nlocal)
    np.add.at(ea, s.owner, out["eatom"])
    cell = ea.reshape(27, n0).sum(axis=1)
    assert np.max(np.abs(cell - cell[0])) < 1e-8 * abs(cell[0])
