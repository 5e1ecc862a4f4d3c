"""Workload scaffolding: everything LAMMPS would hand to ``Pair::compute`` but that
does not exist outside LAMMPS.

None of this is on the hot path.  It stands in for the *callers* of the pair style
(SURVEY.md section 2, rows "Neighbor lists", "Core runtime", "KSpace"): data-file
parsing, ghost-atom construction, the half/newton neighbor list with special-bond bits
(reference: src/neigh_list.h:45-50, src/lmptype.h:58-59), the ``pair_coeff`` mixing
tables (reference: PS.cpp:858-921 ``init_one``) and the 12-bit Coulomb tables
(reference: src/pair.cpp:313-520 ``init_tables``).  Tests, bench.py and smoke() use it
to build inputs in exactly the layout the C-ABI (include/polar_mi355x.h) expects.
"""
from __future__ import annotations

import math
import re
from dataclasses import dataclass, field

import numpy as np

SBBITS = 30
NEIGHMASK = 0x3FFFFFFF
QQR2E_REAL = 332.06371  # reference: src/update.cpp:157 (units real)


# --------------------------------------------------------------------------- data files
def parse_lammps_data(path):
    """Parse an ``atom_style full`` data file (id mol type q x y z) + Bonds section."""
    with open(path) as fh:
        lines = [ln.split("#")[0].strip() for ln in fh]
    box = np.zeros((3, 2))
    natoms = ntypes = nbonds = 0
    for ln in lines[:40]:
        t = ln.split()
        if len(t) == 2 and t[1] == "atoms":
            natoms = int(t[0])
        elif len(t) == 3 and t[1:] == ["atom", "types"]:
            ntypes = int(t[0])
        elif len(t) == 2 and t[1] == "bonds":
            nbonds = int(t[0])
        elif len(t) == 4 and t[2] in ("xlo", "ylo", "zlo"):
            box["xyz".index(t[2][0])] = [float(t[0]), float(t[1])]
    ia = lines.index("Atoms")
    rows = []
    k = ia + 1
    while len(rows) < natoms:
        if lines[k]:
            rows.append(lines[k].split())
        k += 1
    a = np.array([[float(v) for v in r[:7]] for r in rows])
    order = np.argsort(a[:, 0])
    a = a[order]
    bonds = np.zeros((0, 2), dtype=np.int64)
    if nbonds and "Bonds" in lines:
        ib = lines.index("Bonds")
        rows = []
        k = ib + 1
        while len(rows) < nbonds and k < len(lines):
            if lines[k]:
                rows.append(lines[k].split())
            k += 1
        bonds = np.array([[int(r[2]) - 1, int(r[3]) - 1] for r in rows], dtype=np.int64)
    return dict(
        natoms=natoms,
        ntypes=ntypes,
        boxlo=box[:, 0].copy(),
        prd=(box[:, 1] - box[:, 0]).copy(),
        tag=a[:, 0].astype(np.int64),
        molecule=a[:, 1].astype(np.int32),
        type=a[:, 2].astype(np.int32),
        q=a[:, 3].copy(),
        x=a[:, 4:7].copy(),
        bonds=bonds,
    )


def parse_deck(path):
    """Extract the pair-style-relevant numbers from an example input deck
    (``set type T static_polarizability a``, ``pair_style`` args, ``pair_coeff`` rows)."""
    alpha_by_type, coeffs, style_args = {}, [], None
    with open(path) as fh:
        for ln in fh:
            t = ln.split("#")[0].split()
            if not t:
                continue
            if t[0] == "set" and len(t) >= 5 and t[1] == "type" and t[3] == "static_polarizability":
                alpha_by_type[int(t[2])] = float(t[4])
            elif t[0] == "pair_style":
                style_args = t[2:]
            elif t[0] == "pair_coeff":
                coeffs.append(t[1:])
    return dict(alpha_by_type=alpha_by_type, pair_style_args=style_args, pair_coeff=coeffs)


# --------------------------------------------------------------------------- topology
def build_special(n, bonds):
    """1-2 / 1-3 / 1-4 neighbor sets from the bond graph (what LAMMPS' Special computes).
    Returns a dict {(i,j): which} with which in 1..3, both orientations."""
    adj = [[] for _ in range(n)]
    for a, b in bonds:
        adj[a].append(b)
        adj[b].append(a)
    special = {}
    for i in range(n):
        if not adj[i]:
            continue
        d12 = set(adj[i])
        d13 = set()
        for j in d12:
            d13.update(adj[j])
        d13 -= d12 | {i}
        d14 = set()
        for j in d13:
            d14.update(adj[j])
        d14 -= d12 | d13 | {i}
        for j in d12:
            special[(i, j)] = 1
        for j in d13:
            special[(i, j)] = 2
        for j in d14:
            special[(i, j)] = 3
    return special


# --------------------------------------------------------------------------- ghosts + half list
def build_ghosts(x, boxlo, prd, cutghost):
    """Periodic images of local atoms within ``cutghost`` of the box (LAMMPS ghost shell).
    Returns x_all[nall,3], owner[nall] (local index), shift[nall,3] (integer image)."""
    x = np.asarray(x, dtype=np.float64)
    n = len(x)
    xs, owner, shifts = [x], [np.arange(n)], [np.zeros((n, 3), dtype=np.int64)]
    rel = x - boxlo
    nmax = [int(math.ceil(cutghost / prd[k])) for k in range(3)]
    for sx in range(-nmax[0], nmax[0] + 1):
        for sy in range(-nmax[1], nmax[1] + 1):
            for sz in range(-nmax[2], nmax[2] + 1):
                if sx == sy == sz == 0:
                    continue
                s = np.array([sx, sy, sz])
                y = rel + s * prd
                ok = np.all((y >= -cutghost) & (y < prd + cutghost), axis=1)
                if ok.any():
                    idx = np.nonzero(ok)[0]
                    xs.append(x[idx] + s * prd)
                    owner.append(idx)
                    shifts.append(np.tile(s, (len(idx), 1)))
    return np.concatenate(xs), np.concatenate(owner), np.concatenate(shifts)


def build_half_list(x_all, owner, shift, nlocal, cutneigh, molecule=None, special=None,
                    exclude_intra=False, rows=None, full=False, newton=True):
    """Half neighbor list with newton on: every (atom, image) pair within ``cutneigh``
    is stored exactly once, in the list of a LOCAL atom i; j may be a ghost index.
    Special pairs keep their 2-bit code in bits 30-31 (kspace styles keep them in the
    list, reference: src/neighbor.cpp special_flag=2 with a KSpace style).
    ``rows``: optional subset of local atoms that own lists (multi-GPU shards).
    ``full``: LAMMPS *full* list instead (every pair in the rows of both atoms; newton off).
    ``newton=False``: LAMMPS' newton-off HALF list (src/npair_half_bin_newtoff.cpp): a pair is stored by the atom with
    the lower index, ghosts included -- so a local atom lists every ghost partner, and a pair across a periodic face
    appears twice (once per local atom, each with the image of the other)."""
    from scipy.spatial import cKDTree

    tree = cKDTree(x_all)
    loc = np.arange(nlocal) if rows is None else np.asarray(rows)
    numneigh = np.zeros(nlocal, dtype=np.int32)
    chunks = []
    # vectorised over chunks of local atoms
    B = 2048
    for s0 in range(0, len(loc), B):
        ii = loc[s0:s0 + B]
        res = tree.query_ball_point(x_all[ii], cutneigh)
        cnt = np.fromiter((len(r) for r in res), dtype=np.int64, count=len(ii))
        i_rep = np.repeat(ii, cnt)
        j = np.fromiter((v for r in res for v in r), dtype=np.int64, count=int(cnt.sum()))
        oj = owner[j]
        sj = shift[j]
        is_local = j < nlocal
        lex = (sj[:, 0] > 0) | ((sj[:, 0] == 0) & (sj[:, 1] > 0)) | (
            (sj[:, 0] == 0) & (sj[:, 1] == 0) & (sj[:, 2] > 0))
        keep = np.where(is_local, j > i_rep, (oj > i_rep) | ((oj == i_rep) & lex))
        if not newton:
            keep = j > i_rep
        if full:
            keep = j != i_rep
        if exclude_intra and molecule is not None:
            keep &= molecule[i_rep] != molecule[oj]
        i_rep, j, oj = i_rep[keep], j[keep], oj[keep]
        if special:
            code = np.fromiter((special.get((a, b), 0) for a, b in zip(i_rep.tolist(), oj.tolist())),
                               dtype=np.int64, count=len(j))
            j = j | (code << SBBITS)
        order = np.lexsort((j & NEIGHMASK, i_rep))
        i_rep, j = i_rep[order], j[order]
        np.add.at(numneigh, i_rep, 1)
        chunks.append(j.astype(np.int64))
    neigh = np.concatenate(chunks) if chunks else np.zeros(0, dtype=np.int64)
    neigh = neigh.astype(np.uint32).view(np.int32)  # bits 30-31 may be set
    first = np.zeros(nlocal, dtype=np.int64)
    # lists were appended in ascending i within ascending chunks
    csum = np.concatenate([[0], np.cumsum(numneigh[loc])])
    first[loc] = csum[:-1]
    ilist = loc.astype(np.int32)
    return ilist, numneigh, first, neigh


# --------------------------------------------------------------------------- pair tables
def init_one_all(ntypes, coeff_rows, cut_lj_global, cut_coul, mix="geometric", offset_flag=0):
    """pair_coeff rows -> lj1..lj4/offset/cut_ljsq/cutsq tables, reference PS.cpp:772-800 (coeff)
    and PS.cpp:858-921 (init_one); mixing reference src/pair.cpp:660-690.  [(n+1),(n+1)]."""
    w = ntypes + 1
    eps = np.zeros((w, w)); sig = np.zeros((w, w)); cut = np.zeros((w, w))
    setflag = np.zeros((w, w), dtype=np.int32)
    for row in coeff_rows:
        ilo, ihi = _bounds(row[0], ntypes)
        jlo, jhi = _bounds(row[1], ntypes)
        e, s = float(row[2]), float(row[3])
        c = float(row[4]) if len(row) > 4 else cut_lj_global
        for i in range(ilo, ihi + 1):
            for j in range(max(jlo, i), jhi + 1):
                eps[i, j], sig[i, j], cut[i, j], setflag[i, j] = e, s, c, 1
    out = {k: np.zeros((w, w)) for k in ("lj1", "lj2", "lj3", "lj4", "offset", "cut_ljsq", "cutsq")}
    for i in range(1, w):
        for j in range(i, w):
            if not setflag[i, j]:
                e1, e2, s1, s2 = eps[i, i], eps[j, j], sig[i, i], sig[j, j]
                if mix == "geometric":
                    eps[i, j], sig[i, j], cut[i, j] = math.sqrt(e1 * e2), math.sqrt(s1 * s2), math.sqrt(cut[i, i] * cut[j, j])
                elif mix == "arithmetic":
                    eps[i, j], sig[i, j], cut[i, j] = math.sqrt(e1 * e2), 0.5 * (s1 + s2), 0.5 * (cut[i, i] + cut[j, j])
                else:
                    raise ValueError(mix)
            c = max(cut[i, j], cut_coul)
            out["cut_ljsq"][i, j] = cut[i, j] ** 2
            out["lj1"][i, j] = 48.0 * eps[i, j] * sig[i, j] ** 12
            out["lj2"][i, j] = 24.0 * eps[i, j] * sig[i, j] ** 6
            out["lj3"][i, j] = 4.0 * eps[i, j] * sig[i, j] ** 12
            out["lj4"][i, j] = 4.0 * eps[i, j] * sig[i, j] ** 6
            if offset_flag and cut[i, j] > 0:
                r = sig[i, j] / cut[i, j]
                out["offset"][i, j] = 4.0 * eps[i, j] * (r ** 12 - r ** 6)
            out["cutsq"][i, j] = c * c
            for k in out:
                out[k][j, i] = out[k][i, j]
    return out


def _bounds(tok, nmax):
    """reference: Force::bounds wildcard grammar (``*``, ``n*``, ``*n``, ``m*n``)."""
    if "*" not in tok:
        v = int(tok)
        return v, v
    a, b = tok.split("*")
    return (int(a) if a else 1), (int(b) if b else nmax)


def init_bitmap(inner, outer, ntablebits):
    """reference: src/pair.cpp:1676-1723."""
    nlowermin = 1
    while not (2.0 ** nlowermin <= inner * inner and 2.0 ** (nlowermin + 1) > inner * inner):
        nlowermin += 1 if 2.0 ** nlowermin <= inner * inner else -1
    nexpbits = 0
    required, available = outer * outer / 2.0 ** nlowermin, 2.0
    while available < required:
        nexpbits += 1
        available = 2.0 ** (2.0 ** nexpbits)
    nmantbits = ntablebits - nexpbits
    nshift = 24 - (nmantbits + 1)
    nmask = (1 << (ntablebits + nshift)) - 1
    f2i = lambda v: int(np.array([v], dtype=np.float32).view(np.int32)[0])
    return f2i(inner * inner) & ~nmask, f2i(outer * outer) & ~nmask, nmask, nshift


def init_coul_tables(cut_coul, g_ewald, qqrd2e, ncoultablebits=12, tabinner=math.sqrt(2.0)):
    """12-bit bitmapped Coulomb tables, reference src/pair.cpp:313-520 (no rRESPA, no MSM).
    Returns dict(nbits, mask, shift, tabinnersq, tables[8,ntable]) in the order
    r, dr, f, df, c, dc, e, de."""
    from scipy.special import erfc

    masklo, maskhi, nmask, nshift = init_bitmap(tabinner, cut_coul, ncoultablebits)
    ntable = 1 << ncoultablebits
    idx = np.arange(ntable, dtype=np.int64)
    lo = ((idx << nshift) | masklo).astype(np.int32).view(np.float32)
    hi = ((idx << nshift) | maskhi).astype(np.int32).view(np.float32)
    tabinnersq = np.float64(tabinner * tabinner)
    rsq = np.where(lo.astype(np.float64) < tabinnersq, hi, lo).astype(np.float32)
    r = np.sqrt(rsq).astype(np.float64)  # sqrtf
    grij = g_ewald * r
    expm2 = np.exp(-grij * grij)
    derfc = erfc(grij)
    MY_ISPI4 = 1.12837916709551257390
    rt = rsq.astype(np.float64)
    ct = qqrd2e / r
    ft = qqrd2e / r * (derfc + MY_ISPI4 * grij * expm2)
    et = qqrd2e / r * derfc
    minrsq = np.float32(min(np.array([0 | maskhi], dtype=np.int32).view(np.float32)[0], rsq.min()))
    dr = np.empty(ntable); df = np.empty(ntable); dc = np.empty(ntable); de = np.empty(ntable)
    dr[:-1] = 1.0 / (rt[1:] - rt[:-1]); df[:-1] = ft[1:] - ft[:-1]
    dc[:-1] = ct[1:] - ct[:-1]; de[:-1] = et[1:] - et[:-1]
    dr[-1] = 1.0 / (rt[0] - rt[-1]); df[-1] = ft[0] - ft[-1]; dc[-1] = ct[0] - ct[-1]; de[-1] = et[0] - et[-1]
    itablemin = (int(np.array([minrsq]).view(np.int32)[0]) & nmask) >> nshift
    itablemax = itablemin - 1 if itablemin else ntable - 1
    top = np.array([(itablemax << nshift) | maskhi], dtype=np.int32).view(np.float32)[0]
    cut_coulsq = cut_coul * cut_coul
    if float(top) < cut_coulsq:
        rs = np.float32(cut_coulsq)
        rr = float(np.sqrt(rs))
        g = g_ewald * rr
        ex, er = math.exp(-g * g), float(erfc(g))
        dr[itablemax] = 1.0 / (float(rs) - rt[itablemax])
        df[itablemax] = qqrd2e / rr * (er + MY_ISPI4 * g * ex) - ft[itablemax]
        dc[itablemax] = qqrd2e / rr - ct[itablemax]
        de[itablemax] = qqrd2e / rr * er - et[itablemax]
    tables = np.ascontiguousarray(np.stack([rt, dr, ft, df, ct, dc, et, de]))
    return dict(nbits=ncoultablebits, mask=nmask, shift=nshift, tabinnersq=float(minrsq), tables=tables)


def ewald_g(accuracy_rel, q, cutoff, prd, qqrd2e=QQR2E_REAL):
    """Initial g_ewald estimate of ``kspace_style ewald`` (reference: src/KSPACE/ewald.cpp:149-161).
    Out of the hot path: the pair style only reads the resulting scalar (PS.cpp:847)."""
    natoms = len(q)
    two_charge_force = qqrd2e  # force->qqr2e * 1*1/1^2 in units real
    accuracy = accuracy_rel * two_charge_force
    q2 = float(np.sum(q * q)) * qqrd2e
    g = accuracy * math.sqrt(natoms * cutoff * prd[0] * prd[1] * prd[2]) / (2.0 * q2)
    if g >= 1.0:
        return (1.35 - 0.15 * math.log(accuracy)) / cutoff
    return math.sqrt(-math.log(g)) / cutoff


# --------------------------------------------------------------------------- systems
@dataclass
class PolarSettings:
    """pair_style keyword state; defaults are the reference's (PS.cpp:65-78)."""
    cut_lj_global: float = 2.5
    cut_coul: float = 12.0
    iterations_max: int = 50
    damping_type: int = 1  # 0 exponential, 1 none  (reference enum PS.cpp:51)
    polar_damp: float = 2.1304
    zodid: int = 0
    polar_precision: float = 1e-11
    fixed_iteration: int = 0
    polar_gs: int = 0
    polar_gs_ranked: int = 1
    polar_gamma: float = 1.03
    use_previous: int = 0
    debug: int = 0
    dd_cutoff: float = 0.0  # extension: <=0 exact all-pairs (reference), >0 truncated
    device_neigh: int = 0   # extension: the LAMMPS shim builds the LJ/coul list on the device
    restart_polar: int = 0  # extension: restart files carry the polarization keywords
    deterministic: int = 0  # extension: sweeps commit their updates between launches (bit-reproducible runs): 0 = not given (on for fixed_iteration), 1 = yes, 2 = no
    polar_sor: float = 1.0  # extension: over-relaxation factor of the list-mode Gauss-Seidel update (1 = reference)
    rccl_halo: int = 0      # extension: the LAMMPS shim's multi-rank sweeps run through the library's RCCL driver
    polar_accel: int = 0    # extension: Anderson mixing of this depth on the list-mode Gauss-Seidel sweep map (0 = off)


@dataclass
class PolarSystem:
    """One frame in the layout Pair::compute sees (locals first, then ghosts)."""
    nlocal: int
    nghost: int
    x: np.ndarray
    q: np.ndarray
    alpha: np.ndarray
    type: np.ndarray
    molecule: np.ndarray
    boxlo: np.ndarray
    prd: np.ndarray
    ntypes: int
    tables: dict            # lj1..cutsq
    coul: dict              # init_coul_tables output (or nbits=0)
    g_ewald: float
    qqrd2e: float
    special_lj: np.ndarray
    special_coul: np.ndarray
    ilist: np.ndarray
    numneigh: np.ndarray
    firstneigh: np.ndarray
    neigh: np.ndarray
    settings: PolarSettings
    owner: np.ndarray = None
    name: str = ""
    tilt: tuple = (0.0, 0.0, 0.0)   # xy, xz, yz
    triclinic: int = 0
    extra: dict = field(default_factory=dict)


def make_system(x, q, alpha, typ, mol, boxlo, prd, ntypes, coeff_rows, settings, g_ewald,
                bonds=None, exclude_intra=False, skin=2.0, ncoultablebits=12, name="",
                special_lj=(1.0, 0.0, 0.0, 0.0), special_coul=(1.0, 0.0, 0.0, 0.0), rows=None, full=False, newton=True,
                build_list=True):
    """Assemble ghosts, the half list, LJ tables and Coulomb tables for one frame.
    ``build_list=False`` leaves the neighbor list empty: the caller lets the library build it on the device
    (PolarPair.build_neighbors_from_system), which is how the 0.5 M-atom boxes are set up."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    n = len(x)
    tables = init_one_all(ntypes, coeff_rows, settings.cut_lj_global, settings.cut_coul)
    cutmax = math.sqrt(tables["cutsq"][1:, 1:].max())
    cutneigh = cutmax + skin
    x_all, owner, shift = build_ghosts(x, np.asarray(boxlo), np.asarray(prd), cutneigh)
    special = build_special(n, bonds) if bonds is not None and len(bonds) else None
    if build_list:
        ilist, numneigh, first, neigh = build_half_list(
            x_all, owner, shift, n, cutneigh, molecule=np.asarray(mol), special=special,
            exclude_intra=exclude_intra, rows=rows, full=full, newton=newton)
    else:
        ilist, numneigh = np.zeros(0, dtype=np.int32), np.zeros(n, dtype=np.int32)
        first, neigh = np.zeros(n, dtype=np.int64), np.zeros(0, dtype=np.int32)
    coul = init_coul_tables(settings.cut_coul, g_ewald, QQR2E_REAL, ncoultablebits) if ncoultablebits else dict(
        nbits=0, mask=0, shift=0, tabinnersq=0.0, tables=np.zeros((8, 1)))
    g = lambda a, dt: np.ascontiguousarray(np.asarray(a)[owner], dtype=dt)
    return PolarSystem(
        nlocal=n, nghost=len(x_all) - n, x=np.ascontiguousarray(x_all), q=g(q, np.float64),
        alpha=g(alpha, np.float64), type=g(typ, np.int32), molecule=g(mol, np.int32),
        boxlo=np.asarray(boxlo, dtype=np.float64), prd=np.asarray(prd, dtype=np.float64), ntypes=ntypes,
        tables=tables, coul=coul, g_ewald=g_ewald, qqrd2e=QQR2E_REAL,
        special_lj=np.asarray(special_lj, dtype=np.float64), special_coul=np.asarray(special_coul, dtype=np.float64),
        ilist=ilist, numneigh=numneigh, firstneigh=first, neigh=neigh, settings=settings, owner=owner, name=name,
        extra={"coeff_rows": [list(map(str, r)) for r in coeff_rows], "special": special, "cutneigh": cutneigh,
               "exclude_intra": bool(exclude_intra), "newton_pair": int(bool(newton))})


def compact_shard(s, own, halo):
    """The part of a replicated system ONE rank needs: its own atoms (rows), the halo atoms other
    ranks own but its rows can see, and the ghost images its LJ/Coulomb rows reference -- in that
    order, with the neighbor rows of ``own`` re-indexed.  The global box is kept (minimum image
    still resolves the periodic wrap between the first and the last slab).
    ``s`` must carry list rows for ``own`` (replicate_fixture(rows=own, full=True))."""
    own = np.asarray(own, dtype=np.int64)
    halo = np.asarray(halo, dtype=np.int64)
    n, nall = s.nlocal, s.nlocal + s.nghost
    seg = [(int(s.firstneigh[i]), int(s.numneigh[i])) for i in own]
    flat = np.concatenate([s.neigh[a:a + c] for a, c in seg]) if seg else np.zeros(0, dtype=np.int32)
    u = flat.view(np.uint32)
    jm = (u & np.uint32(NEIGHMASK)).astype(np.int64)
    bits = u & np.uint32(0xC0000000)
    ghosts = np.unique(jm[jm >= n])
    lut = np.full(nall, -1, dtype=np.int64)
    lut[own] = np.arange(len(own))
    lut[halo] = len(own) + np.arange(len(halo))
    lut[ghosts] = len(own) + len(halo) + np.arange(len(ghosts))
    jc = lut[jm]
    if np.any(jc < 0):
        raise ValueError("a neighbor of an owned row is neither owned, halo nor ghost: halo reach too small")
    neigh = (jc.astype(np.uint32) | bits).view(np.int32)
    keep = np.concatenate([own, halo, ghosts])
    nloc = len(own) + len(halo)
    numneigh = np.zeros(nloc, dtype=np.int32)
    numneigh[:len(own)] = [c for _, c in seg]
    first = np.zeros(nloc, dtype=np.int64)
    first[:len(own)] = np.concatenate([[0], np.cumsum(numneigh[:len(own)])])[:-1]
    return PolarSystem(
        nlocal=nloc, nghost=len(ghosts), x=np.ascontiguousarray(s.x[keep]), q=s.q[keep].copy(), alpha=s.alpha[keep].copy(),
        type=s.type[keep].copy(), molecule=s.molecule[keep].copy(), boxlo=s.boxlo, prd=s.prd, ntypes=s.ntypes,
        tables=s.tables, coul=s.coul, g_ewald=s.g_ewald, qqrd2e=s.qqrd2e, special_lj=s.special_lj,
        special_coul=s.special_coul, ilist=np.arange(len(own), dtype=np.int32), numneigh=numneigh, firstneigh=first,
        neigh=neigh, settings=s.settings, owner=None, name=s.name + "_shard", extra=dict(s.extra, n_own=len(own)))


def compact_shard_geometric(s, own, halo, reach):
    """compact_shard for systems made WITHOUT a host neighbor list (``build_list=False``; the library builds the list
    of the own rows on the device): [own | halo | ghosts], ghosts = the periodic images within ``reach`` of the
    bounding box of the own atoms (a superset of what the own rows can reference).  ``owner`` keeps the global atom
    ids (they serve as tags in polar_build_neighbors)."""
    own = np.asarray(own, dtype=np.int64)
    halo = np.asarray(halo, dtype=np.int64)
    n, nall = s.nlocal, s.nlocal + s.nghost
    lo = s.x[own].min(axis=0) - reach
    hi = s.x[own].max(axis=0) + reach
    g = np.arange(n, nall)
    xg = s.x[g]
    ghosts = g[np.all((xg >= lo) & (xg <= hi), axis=1)]
    keep = np.concatenate([own, halo, ghosts])
    nloc = len(own) + len(halo)
    return PolarSystem(
        nlocal=nloc, nghost=len(ghosts), x=np.ascontiguousarray(s.x[keep]), q=s.q[keep].copy(), alpha=s.alpha[keep].copy(),
        type=s.type[keep].copy(), molecule=s.molecule[keep].copy(), boxlo=s.boxlo, prd=s.prd, ntypes=s.ntypes,
        tables=s.tables, coul=s.coul, g_ewald=s.g_ewald, qqrd2e=s.qqrd2e, special_lj=s.special_lj,
        special_coul=s.special_coul, ilist=np.zeros(0, dtype=np.int32), numneigh=np.zeros(nloc, dtype=np.int32),
        firstneigh=np.zeros(nloc, dtype=np.int64), neigh=np.zeros(0, dtype=np.int32), settings=s.settings,
        owner=np.asarray(s.owner)[keep], name=s.name + "_shard", extra=dict(s.extra, n_own=len(own)))


def slab_order(s, axis=2, small=16, glue_dist=0.0, max_cluster=64):
    """(order, key, glue) for a slab decomposition along ``axis``: the local atoms by coordinate, with groups of atoms that
    must not end up on two ranks -- where they would see each other's dipoles one sweep late, Jacobi-fashion -- moving as one
    (all members get the coordinate of the group's first atom):
      * a small molecule (<= ``small`` atoms: a rigid sorbate, sites a fraction of an angstrom apart);
      * with ``glue_dist`` > 0, clusters of atoms connected by distances below it (bonded neighbours of a framework: the
        strongest couplings of the dipole field matrix), up to ``max_cluster`` atoms and not reaching around the periodic box.
    Stable; ``glue[k]``: sorted atom k belongs to the same group as sorted atom k - 1 (for split_sorted)."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components

    n = s.nlocal
    key = np.array(s.x[:n, axis], dtype=np.float64)
    mol = np.asarray(s.molecule[:n])
    rows, cols = [], []
    if len(mol) and mol.max() > 0:   # consecutive sites of one small molecule
        counts = np.bincount(mol)
        idx = np.nonzero((mol > 0) & (counts[mol] <= small))[0]
        by_mol = idx[np.argsort(mol[idx], kind="stable")]
        same = mol[by_mol[1:]] == mol[by_mol[:-1]]
        rows.append(by_mol[:-1][same]); cols.append(by_mol[1:][same])
    if glue_dist > 0.0:
        from scipy.spatial import cKDTree
        prd = np.asarray(s.prd, dtype=np.float64)
        xw = np.mod(np.asarray(s.x[:n]) - np.asarray(s.boxlo), prd)
        xw = np.where(xw >= prd, 0.0, xw)
        pairs = cKDTree(xw, boxsize=prd).query_pairs(glue_dist, output_type="ndarray")
        rows.append(pairs[:, 0]); cols.append(pairs[:, 1])
    group = np.arange(n)
    if rows and sum(len(r) for r in rows):
        r, c = np.concatenate(rows), np.concatenate(cols)
        _, lab = connected_components(coo_matrix((np.ones(len(r)), (r, c)), shape=(n, n)), directed=False)
        size = np.bincount(lab)
        zmin = np.full(len(size), np.inf); zmax = np.full(len(size), -np.inf)
        np.minimum.at(zmin, lab, key); np.maximum.at(zmax, lab, key)
        first = np.full(len(size), n, dtype=np.int64)
        np.minimum.at(first, lab, np.arange(n))
        ok = (size > 1) & (size <= max(max_cluster, small)) & (zmax - zmin < 0.5 * float(s.prd[axis]))
        member = ok[lab]
        key[member] = key[first[lab[member]]]
        group = np.where(member, first[lab], np.arange(n))
    order = np.argsort(key, kind="stable")
    # members of a group share a key; make them adjacent even when another atom happens to have exactly that coordinate
    order = order[np.lexsort((group[order], key[order]))]
    go = group[order]
    glue = np.zeros(n, dtype=bool)
    if n > 1:
        glue[1:] = go[1:] == go[:-1]
    return order, key, glue


def brick_order(s, grid, small=16, glue_dist=0.0, max_cluster=64):
    """(order, offs) for a brick decomposition grid = (gx, gy, gz): recursive coordinate bisection into equal atom counts --
    gz slabs along z, each cut into gy along y, each of those into gx along x -- with the same groups kept whole as
    slab_order (a group moves with the coordinates of its first atom).  Rank r = (iz * gy + iy) * gx + ix owns
    order[offs[r]:offs[r+1]]."""
    gx, gy, gz = (int(v) for v in grid)
    n = s.nlocal
    # group representative coordinates along every axis (slab_order computes the groups once per axis; they are the same)
    keys, glues = [], None
    for axis in range(3):
        o, key, glue = slab_order(s, axis=axis, small=small, glue_dist=glue_dist, max_cluster=max_cluster)
        keys.append(key)
    # group id = index of the first atom with the same representative triple (members share all three keys)
    rep = np.stack(keys, axis=1)
    _, group = np.unique(rep, axis=0, return_inverse=True)
    group = group.reshape(-1)

    def split(idx, axis, parts):
        """idx (atom indices) into `parts` pieces of equal count along `axis`, groups whole"""
        k = keys[axis][idx]
        o = np.lexsort((group[idx], k))
        idx = idx[o]
        g = group[idx]
        glue = np.zeros(len(idx), dtype=bool)
        if len(idx) > 1:
            glue[1:] = (g[1:] == g[:-1]) & (k[o][1:] == k[o][:-1])
        _, offs = split_sorted(k[o], parts, glue)
        return [idx[offs[r]:offs[r + 1]] for r in range(parts)]

    pieces = []
    for zs in split(np.arange(n), 2, gz):
        for ys in split(zs, 1, gy):
            pieces.extend(split(ys, 0, gx))
    order = np.concatenate(pieces) if pieces else np.zeros(0, dtype=np.int64)
    offs = np.concatenate([[0], np.cumsum([len(q) for q in pieces])]).astype(int)
    return order, offs


def split_sorted(key_sorted, world, glue=None):
    """Equal-count split of a sorted key into ``world`` contiguous ranges.  ``glue[k]`` (optional) says that element k belongs
    with element k - 1 (two sites of one small molecule): a boundary that would fall between them moves up."""
    n = len(key_sorted)
    offs = [0]
    for r in range(1, world):
        o = max((r * n) // world, offs[-1])
        while glue is not None and 0 < o < n and glue[o]:
            o += 1
        offs.append(o)
    offs.append(n)
    offs = np.asarray(offs, dtype=int)
    return [int(offs[r + 1] - offs[r]) for r in range(world)], offs


def permute_locals(s, order):
    """The same system with its local atoms in the order ``order`` (systems made with ``build_list=False`` only: there is no
    neighbor list to re-index).  Ghosts keep their places; their ``owner`` entries follow the locals."""
    if len(s.neigh) or s.extra.get("special"):
        raise ValueError("permute_locals: the system carries lists indexed by atom")
    n, nall = s.nlocal, s.nlocal + s.nghost
    order = np.asarray(order, dtype=np.int64)
    inv = np.empty(n, dtype=np.int64)
    inv[order] = np.arange(n)
    full = np.concatenate([order, np.arange(n, nall)])
    owner = np.asarray(s.owner)
    new_owner = inv[owner[full]]
    return PolarSystem(
        nlocal=n, nghost=s.nghost, x=np.ascontiguousarray(s.x[full]), q=s.q[full].copy(), alpha=s.alpha[full].copy(),
        type=s.type[full].copy(), molecule=s.molecule[full].copy(), boxlo=s.boxlo, prd=s.prd, ntypes=s.ntypes,
        tables=s.tables, coul=s.coul, g_ewald=s.g_ewald, qqrd2e=s.qqrd2e, special_lj=s.special_lj,
        special_coul=s.special_coul, ilist=s.ilist, numneigh=s.numneigh, firstneigh=s.firstneigh, neigh=s.neigh,
        settings=s.settings, owner=new_owner, name=s.name, tilt=s.tilt, triclinic=s.triclinic, extra=dict(s.extra))


def lammps_special_arrays(n, special):
    """{(i,j): which} -> atom->nspecial [n][3] (cumulative 1-2, 1-3, 1-4 counts) and atom->special
    [n][maxspecial] (partner TAGS = index + 1, ordered 1-2 | 1-3 | 1-4), the layout
    NPair::find_special reads (reference src/npair.cpp)."""
    per = [[[], [], []] for _ in range(n)]
    for (i, j), which in (special or {}).items():
        per[i][which - 1].append(j + 1)
    maxspecial = max([sum(len(c) for c in p) for p in per] + [1])
    nspecial = np.zeros((n, 3), dtype=np.int32)
    arr = np.zeros((n, maxspecial), dtype=np.int32)
    for i, p in enumerate(per):
        flat = sorted(p[0]) + sorted(p[1]) + sorted(p[2])
        arr[i, :len(flat)] = flat
        nspecial[i] = np.cumsum([len(p[0]), len(p[1]), len(p[2])])
    return nspecial, arr


def neighbor_special_flag(special_lj, special_coul, kspace=True):
    """neighbor->special_flag (reference src/neighbor.cpp init): 0 drop, 1 keep plain, 2 keep with bits;
    with a KSpace style every special pair is kept with its bits."""
    flag = [1, 0, 0, 0]
    for k in (1, 2, 3):
        if special_lj[k] == 0.0 and special_coul[k] == 0.0:
            flag[k] = 0
        elif special_lj[k] == 1.0 and special_coul[k] == 1.0:
            flag[k] = 1
        else:
            flag[k] = 2
        if kspace:
            flag[k] = 2
    return np.asarray(flag, dtype=np.int32)


# --------------------------------------------------------------------------- synthetic boxes
# MOF5+H2 per-type tables (values as quoted in SURVEY.md section 8(d); polarizabilities in A^3)
_MOF_ALPHA = np.array([0.16, 0.852, 0.852, 0.4138, 1.2886, 1.2886, 1.2886])
_MOF_Q = np.array([1.853, -1.0069, -2.2568, 0.1489, 1.0983, -0.0518, -0.1378])
_MOF_FRAC = np.array([32, 96, 8, 96, 48, 96, 48], dtype=np.float64) / 424.0
_H2_SITES = np.array([[0.0, 0.0, 0.0], [-0.371, 0.0, 0.0], [0.371, 0.0, 0.0], [-0.363, 0.0, 0.0], [0.363, 0.0, 0.0]])
_H2_Q = np.array([-0.7464, 0.3732, 0.3732, 0.0, 0.0])
_H2_ALPHA = np.array([0.6938, 0.00044, 0.00044, 0.0, 0.0])
RHO = 0.0798  # atoms / A^3 (MOF5+H2 example density)

# LJ table of the synthetic box: ten types, Lorentz-Berthelot-like, cut 2.5 sigma
_SYN_EPS = np.array([0.12398, 0.059984, 0.059984, 0.043989, 0.104987, 0.104987, 0.104987, 0.025363, 0.0, 0.004306])
_SYN_SIG = np.array([2.462, 3.118, 3.118, 2.571, 3.431, 3.431, 3.431, 3.15528, 0.0, 2.37031])


def synth_coeff_rows():
    rows = []
    for i in range(10):
        for j in range(i, 10):
            e = math.sqrt(_SYN_EPS[i] * _SYN_EPS[j])
            s = 0.5 * (_SYN_SIG[i] + _SYN_SIG[j])
            rows.append([str(i + 1), str(j + 1), f"{e:.6f}", f"{s:.6f}", f"{2.5 * s:.6f}"])
    return rows


def synth(N, seed=1):
    """Synthetic polarizable MOF+sorbate box (SURVEY.md section 8(d) generator):
    31 % "framework" atoms on a jittered simple-cubic sub-lattice (molecule id 1, types 1-7),
    69 % rigid 5-site H2 molecules at random centres/orientations.  Returns a dict of
    per-atom arrays and the cubic box."""
    rng = np.random.default_rng(seed)
    L = (N / RHO) ** (1.0 / 3.0)
    nmol = int(round(0.69 * N / 5.0))
    nsorb = 5 * nmol
    nfw = N - nsorb
    # framework: jittered simple-cubic sub-lattice
    m = int(math.ceil(nfw ** (1.0 / 3.0)))
    a = L / m
    grid = np.stack(np.meshgrid(np.arange(m), np.arange(m), np.arange(m), indexing="ij"), -1).reshape(-1, 3)
    pick = rng.permutation(len(grid))[:nfw]
    xfw = (grid[pick] + 0.5) * a + rng.normal(0.0, 0.1, (nfw, 3))
    tfw = rng.choice(7, size=nfw, p=_MOF_FRAC / _MOF_FRAC.sum())
    qfw = _MOF_Q[tfw].copy()
    qfw -= qfw.mean()
    # sorbate: random centres with a minimum separation to the framework sites and each other
    from scipy.spatial import cKDTree
    centres = np.zeros((0, 3))
    tree_fw = cKDTree(np.mod(xfw, L), boxsize=L)
    while len(centres) < nmol:
        cand = rng.uniform(0.0, L, (2 * (nmol - len(centres)) + 64, 3))
        d, _ = tree_fw.query(cand)
        cand = cand[d > 1.6]
        allc = np.concatenate([centres, cand])
        t = cKDTree(allc, boxsize=L)
        bad = set()
        for i, j in t.query_pairs(2.5):
            bad.add(max(i, j))
        keep = np.array([k for k in range(len(allc)) if k not in bad], dtype=np.int64)
        centres = allc[keep][:nmol]
    # random rotations
    v = rng.normal(size=(nmol, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
    xs = centres[:, None, :] + _H2_SITES[None, :, 0, None] * v[:, None, :]
    x = np.concatenate([xfw, xs.reshape(-1, 3)])
    x = np.mod(x, L)
    typ = np.concatenate([tfw + 1, np.tile(np.array([8, 9, 9, 10, 10]), nmol)]).astype(np.int32)
    q = np.concatenate([qfw, np.tile(_H2_Q, nmol)])
    alpha = np.concatenate([_MOF_ALPHA[tfw], np.tile(_H2_ALPHA, nmol)])
    mol = np.concatenate([np.ones(nfw, dtype=np.int32), np.repeat(np.arange(2, nmol + 2, dtype=np.int32), 5)])
    return dict(N=N, L=L, x=x, q=q, alpha=alpha, type=typ, molecule=mol,
                boxlo=np.zeros(3), prd=np.array([L, L, L]), ntypes=10)


# --------------------------------------------------------------------------- fixtures
def parse_pair_style_args(args, base=None):
    """Python mirror of the reference keyword scan (PS.cpp:678-766) used to turn fixture
    metadata into PolarSettings.  The product parser is the C++ one behind
    ``polar_pair_settings`` (csrc/pair_host.cpp); tests check the two agree."""
    st = base or PolarSettings()
    if len(args) < 1:
        raise ValueError("Illegal pair_style command")
    st.cut_lj_global = float(args[0])
    st.cut_coul = st.cut_lj_global if len(args) == 1 else float(args[1])
    i = 2
    yn = {"yes": 1, "no": 0}
    while i < len(args):
        if i + 2 > len(args):
            raise ValueError("Illegal pair_style command")
        k, v = args[i], args[i + 1]
        if k == "precision":
            st.polar_precision = float(v)
        elif k == "zodid":
            if st.polar_gs or st.polar_gs_ranked:
                raise ValueError("Zodid doesn't work with polar_gs or polar_gs_ranked")
            st.zodid = yn[v]
        elif k == "fixed_iteration":
            st.fixed_iteration = yn[v]
        elif k == "damp":
            st.polar_damp = float(v)
        elif k == "max_iterations":
            st.iterations_max = int(v)
        elif k == "damp_type":
            st.damping_type = {"exponential": 0, "none": 1}[v]
        elif k == "polar_gs":
            if st.polar_gs_ranked:
                raise ValueError("polar_gs and polar_gs_ranked are mutually exclusive")
            st.polar_gs = yn[v]
        elif k == "polar_gs_ranked":
            if st.polar_gs:
                raise ValueError("polar_gs and polar_gs_ranked are mutually exclusive")
            st.polar_gs_ranked = yn[v]
        elif k == "polar_gamma":
            st.polar_gamma = float(v)
        elif k == "debug":
            st.debug = yn[v]
        elif k == "use_previous":
            st.use_previous = yn[v]
        elif k == "dd_cutoff":  # extension keyword (not in the reference)
            st.dd_cutoff = float(v)
        elif k == "device_neigh":  # extension keyword (not in the reference)
            st.device_neigh = yn[v]
        elif k == "restart_polar":  # extension keyword (not in the reference)
            st.restart_polar = yn[v]
        elif k == "deterministic":  # extension keyword (not in the reference)
            st.deterministic = 1 if yn[v] else 2   # POLAR_DET_YES / POLAR_DET_NO (0 = keyword not given: on for fixed_iteration runs)
        elif k == "rccl_halo":  # extension keyword (not in the reference)
            st.rccl_halo = yn[v]
        elif k == "polar_accel":  # extension keyword (not in the reference)
            st.polar_accel = int(v)
            if not 0 <= st.polar_accel <= 8:
                raise ValueError("Illegal pair_style command")
        elif k == "polar_sor":  # extension keyword (not in the reference)
            st.polar_sor = float(v)
            if not 0.0 < st.polar_sor < 2.0:
                raise ValueError("Illegal pair_style command")
        else:
            raise ValueError("Illegal pair_style command")
        i += 2
    return st


def load_fixture(path, extra_args=(), g_ewald=None, ncoultablebits=12, newton=True):
    """tests/golden/<case>.npz -> (PolarSystem, meta dict)."""
    import json

    z = np.load(path)
    meta = json.loads(str(z["meta"]))
    st = parse_pair_style_args(list(meta["pair_style_args"]) + list(extra_args))
    rows = [[str(int(r[0])), str(int(r[1])), repr(float(r[2])), repr(float(r[3])), repr(float(r[4]))]
            for r in z["pair_coeff"]]
    g = meta["known"]["g_ewald"] if g_ewald is None else g_ewald
    sysm = make_system(z["x"], z["q"], z["alpha"], z["type"], z["molecule"], z["boxlo"], z["prd"],
                       meta["ntypes"], rows, st, g, bonds=z["bonds"], exclude_intra=meta["exclude_intra"],
                       ncoultablebits=ncoultablebits, name=meta["name"], newton=newton)
    return sysm, meta


def synth_system(N, seed=1, extra_args=(), cut_lj=2.5, cut_coul=12.8345, skin=2.0, build_list=True):
    """PolarSystem of the synthetic generator (SURVEY.md 8(d) "primary" configs 1-4): synth(N, seed)
    with the MOF5+H2 deck's pair_style line (exponential damping a = 2.1304, ranked GS) plus ``extra_args``.

    A LOAD GENERATOR, not a physical system: sorbate centres are placed without regard to the framework sites, so
    sites overlap (E_pol ~ -2 kcal/mol per atom, 350x the MOF replica) and the solver's convergence history means
    nothing.  It has the right density, list lengths and memory pattern; bench.py uses it only behind ``--synth`` and the
    headline and every parity property run on the replicated MOF5+H2 cell."""
    d = synth(N, seed)
    args = [repr(cut_lj), repr(cut_coul), "damp_type", "exponential", "damp", "2.1304", "polar_gs_ranked", "yes",
            "use_previous", "no"] + list(extra_args)
    st = parse_pair_style_args(args)
    g = ewald_g(1.0e-4, d["q"], st.cut_coul, d["prd"])
    return make_system(d["x"], d["q"], d["alpha"], d["type"], d["molecule"], d["boxlo"], d["prd"], d["ntypes"],
                       synth_coeff_rows(), st, g, bonds=None, exclude_intra=True, skin=skin, name=f"synth{N}_s{seed}", build_list=build_list)


def replicate_fixture(path, nx, ny, nz, extra_args=(), g_ewald=None, skin=2.0, rows=None, full=False, build_list=True):
    """``replicate nx ny nz`` of a golden fixture (BASELINE configs[1..4] are replicated boxes).
    Follows LAMMPS' replicate semantics: molecule ids are offset per replica (so framework
    replicas no longer exclude each other, SURVEY.md 8(d) caveat); bonds are not replicated
    (intramolecular pairs are removed by ``neigh_modify exclude molecule/intra`` anyway)."""
    import json

    z = np.load(path)
    meta = json.loads(str(z["meta"]))
    st = parse_pair_style_args(list(meta["pair_style_args"]) + list(extra_args))
    x0, prd0 = z["x"], z["prd"]
    n0 = len(x0)
    molmax = int(z["molecule"].max())
    xs, mols = [], []
    r = 0
    for iz in range(nz):          # z outermost: contiguous index ranges are z slabs
        for iy in range(ny):
            for ix in range(nx):
                xs.append(x0 + np.array([ix, iy, iz]) * prd0)
                mols.append(z["molecule"] + r * molmax)
                r += 1
    nrep = nx * ny * nz
    x = np.concatenate(xs)
    coeff_rows = [[str(int(c[0])), str(int(c[1])), repr(float(c[2])), repr(float(c[3])), repr(float(c[4]))]
                  for c in z["pair_coeff"]]
    prd = prd0 * np.array([nx, ny, nz])
    q = np.tile(z["q"], nrep)
    g = ewald_g(1.0e-4, q, st.cut_coul, prd) if g_ewald is None else g_ewald
    return make_system(x, q, np.tile(z["alpha"], nrep), np.tile(z["type"], nrep), np.concatenate(mols),
                       z["boxlo"], prd, meta["ntypes"], coeff_rows, st, g, bonds=None, exclude_intra=True, skin=skin,
                       name=f"{meta['name']}_rep{nx}x{ny}x{nz}", rows=rows, full=full, build_list=build_list)
