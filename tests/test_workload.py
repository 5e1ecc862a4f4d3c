"""CPU tests of the input scaffolding that only the multi-GPU bench uses (workload.slab_order / split_sorted / permute_locals)."""
import numpy as np
import pytest


def test_slab_order_keeps_molecules_whole_and_permutation_consistent(wl):
    """Geometric slabs for the multi-GPU bench: locals sorted along z, a sorbate molecule as one; equal-count split that never
    separates equal keys; the permuted system is the same system (ghosts still point at their owners)."""
    import os
    from helpers import GOLD
    s = wl.replicate_fixture(os.path.join(GOLD, "mof5_h2.npz"), 1, 1, 3, extra_args=["dd_cutoff", "12.8345"], build_list=False)
    order, key, glue = wl.slab_order(s, axis=2)
    assert sorted(order.tolist()) == list(range(s.nlocal)) and np.all(np.diff(key[order]) >= 0)
    counts, offs = wl.split_sorted(key[order], 4, glue)
    assert sum(counts) == s.nlocal and max(counts) - min(counts) <= 16
    sp = wl.permute_locals(s, order)
    n = s.nlocal
    assert np.array_equal(sp.x[:n], s.x[order]) and np.array_equal(sp.x[n:], s.x[n:])
    g = np.arange(n, n + s.nghost)
    for name in ("q", "alpha", "type", "molecule"):
        assert np.array_equal(getattr(sp, name)[g], getattr(sp, name)[sp.owner[g]]), name
    d = sp.x[g] - sp.x[sp.owner[g]]                       # a ghost is its owner shifted by whole box vectors
    assert np.allclose(d / s.prd, np.round(d / s.prd), atol=1e-9)
    mol = sp.molecule[:n]
    cnt = np.bincount(mol)
    for r in range(4):
        a, b = offs[r], offs[r + 1]
        for m in np.unique(mol[a:b]):
            if m > 0 and cnt[m] <= 16:
                assert np.count_nonzero(mol[a:b] == m) == cnt[m]     # the whole molecule is on this rank
    # clusters of bonded framework atoms (< 1.6 A) stay whole as well
    order2, key2, glue2 = wl.slab_order(s, axis=2, glue_dist=1.6)
    from scipy.spatial import cKDTree
    xw = np.mod(s.x[:n] - s.boxlo, s.prd)
    pairs = cKDTree(xw, boxsize=s.prd).query_pairs(1.6, output_type="ndarray")
    pos = np.empty(n, dtype=int); pos[order2] = np.arange(n)
    c2, o2 = wl.split_sorted(key2[order2], 4, glue2)
    rank_of = np.searchsorted(o2[1:], pos, side="right")
    split = rank_of[pairs[:, 0]] != rank_of[pairs[:, 1]]
    zext = np.abs(s.x[pairs[:, 0], 2] - s.x[pairs[:, 1], 2])
    assert np.count_nonzero(split & (zext < 2.0)) == 0       # only pairs that reach around the periodic box may be cut
    assert max(c2) - min(c2) <= 64
    with pytest.raises(ValueError):
        wl.permute_locals(wl.replicate_fixture(os.path.join(GOLD, "mof5_h2.npz"), 1, 1, 1), np.arange(1349))


def test_brick_order_partitions_into_equal_bricks_with_molecules_whole(wl):
    import os
    from helpers import GOLD
    s = wl.replicate_fixture(os.path.join(GOLD, "mof5_h2.npz"), 2, 2, 2, extra_args=["dd_cutoff", "12.8345"], build_list=False)
    order, offs = wl.brick_order(s, (2, 2, 2), glue_dist=1.6)
    n = s.nlocal
    assert sorted(order.tolist()) == list(range(n)) and len(offs) == 9 and offs[-1] == n
    counts = np.diff(offs)
    assert counts.max() - counts.min() <= 64
    rank = np.empty(n, dtype=int)
    for r in range(8):
        rank[order[offs[r]:offs[r + 1]]] = r
    mol = s.molecule[:n]
    cnt = np.bincount(mol)
    for m in np.unique(mol[(mol > 0) & (cnt[mol] <= 16)])[::7]:
        assert len(set(rank[mol == m])) == 1
    # rank r = (iz * 2 + iy) * 2 + ix: the bricks are ordered along x fastest
    cx = [np.median(s.x[order[offs[r]:offs[r + 1]], 0]) for r in range(8)]
    cz = [np.median(s.x[order[offs[r]:offs[r + 1]], 2]) for r in range(8)]
    assert cx[0] < cx[1] and cx[2] < cx[3] and max(cz[:4]) < min(cz[4:])
