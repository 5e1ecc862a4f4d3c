"""ctypes binding of the CPU oracle (oracle/polar_oracle.c).

TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this module; the product path never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int)
llp = C.POINTER(C.c_longlong)


class OrcSystem(C.Structure):
    _fields_ = [
        ("nlocal", C.c_int), ("nghost", C.c_int),
        ("x", dp), ("q", dp), ("alpha", dp), ("type", ip), ("molecule", ip),
        ("prd", C.c_double * 3), ("tilt", C.c_double * 3), ("periodic", C.c_int * 3), ("triclinic", C.c_int),
        ("ntypes", C.c_int),
        ("lj1", dp), ("lj2", dp), ("lj3", dp), ("lj4", dp), ("offset", dp), ("cut_ljsq", dp), ("cutsq", dp),
        ("cut_coul", C.c_double), ("g_ewald", C.c_double), ("qqrd2e", C.c_double),
        ("special_lj", C.c_double * 4), ("special_coul", C.c_double * 4),
        ("newton_pair", C.c_int),
        ("ncoultablebits", C.c_int), ("ncoulmask", C.c_int), ("ncoulshiftbits", C.c_int),
        ("tabinnersq", C.c_double),
        ("rtable", dp), ("drtable", dp), ("ftable", dp), ("dftable", dp),
        ("ctable", dp), ("dctable", dp), ("etable", dp), ("detable", dp),
        ("inum", C.c_int), ("ilist", ip), ("numneigh", ip), ("firstneigh", llp), ("neigh", ip),
        ("iterations_max", C.c_int), ("damping_type", C.c_int), ("zodid", C.c_int), ("fixed_iteration", C.c_int),
        ("polar_gs", C.c_int), ("polar_gs_ranked", C.c_int), ("use_previous", C.c_int), ("debug", C.c_int),
        ("polar_damp", C.c_double), ("polar_precision", C.c_double), ("polar_gamma", C.c_double),
        ("dd_cutoff", C.c_double),
    ]


class OrcResult(C.Structure):
    _fields_ = [
        ("eng_vdwl", C.c_double), ("eng_coul", C.c_double), ("eng_pol", C.c_double),
        ("u_self", C.c_double), ("u_ef", C.c_double), ("u_dd", C.c_double),
        ("virial", C.c_double * 6), ("rmin", C.c_double),
        ("iterations", C.c_int), ("status", C.c_int), ("rms_dmu", C.c_double),
        ("t_rank", C.c_double), ("t_ljcoul", C.c_double), ("t_static", C.c_double),
        ("t_matrix", C.c_double), ("t_solve", C.c_double), ("t_force", C.c_double),
        ("sweeps", C.c_int),
        ("force_atom0", C.c_double * 3), ("dipole_force_atom0", C.c_double * 3),
    ]


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "polar_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(os.environ.get("POLAR_ORACLE_SO") or build())   # the override: a sanitizer build (tests/test_sanitizers.py)
        _LIB.orc_compute.restype = C.c_int
        _LIB.orc_compute.argtypes = [C.POINTER(OrcSystem), C.c_int, C.c_int, dp, dp, dp, C.POINTER(OrcResult), dp]
        _LIB.orc_compute_peratom.restype = C.c_int
        _LIB.orc_compute_peratom.argtypes = _LIB.orc_compute.argtypes + [dp, dp]
        _LIB.orc_init_tables.restype = C.c_int
        _LIB.orc_init_tables.argtypes = [C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, ip, ip, dp, dp]
        _LIB.orc_rank_metric.argtypes = [C.POINTER(OrcSystem), dp, dp]
        _LIB.orc_static_field.argtypes = [C.POINTER(OrcSystem), dp]
    return _LIB


def _p(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


def make_struct(sys, settings=None):
    """PolarSystem (workload.py) -> (OrcSystem, keepalive list)."""
    st = settings or sys.settings
    keep = []

    def arr(a, dt):
        b = np.ascontiguousarray(a, dtype=dt)
        keep.append(b)
        return b

    s = OrcSystem()
    s.nlocal, s.nghost = sys.nlocal, sys.nghost
    s.x = _p(arr(sys.x, np.float64), C.c_double)
    s.q = _p(arr(sys.q, np.float64), C.c_double)
    s.alpha = _p(arr(sys.alpha, np.float64), C.c_double)
    s.type = _p(arr(sys.type, np.int32), C.c_int)
    s.molecule = _p(arr(sys.molecule, np.int32), C.c_int)
    s.prd[:] = list(sys.prd)
    s.tilt[:] = list(getattr(sys, "tilt", (0.0, 0.0, 0.0)))
    s.periodic[:] = [1, 1, 1]
    s.triclinic = int(getattr(sys, "triclinic", 0))
    s.ntypes = sys.ntypes
    for k in ("lj1", "lj2", "lj3", "lj4", "offset", "cut_ljsq", "cutsq"):
        setattr(s, k, _p(arr(sys.tables[k], np.float64), C.c_double))
    s.cut_coul, s.g_ewald, s.qqrd2e = st.cut_coul, sys.g_ewald, sys.qqrd2e
    s.special_lj[:] = list(sys.special_lj)
    s.special_coul[:] = list(sys.special_coul)
    s.newton_pair = int(getattr(sys, "extra", {}).get("newton_pair", 1))
    s.ncoultablebits = sys.coul["nbits"]
    s.ncoulmask, s.ncoulshiftbits = sys.coul["mask"], sys.coul["shift"]
    s.tabinnersq = sys.coul["tabinnersq"]
    tb = arr(sys.coul["tables"], np.float64)
    for k, name in enumerate(("rtable", "drtable", "ftable", "dftable", "ctable", "dctable", "etable", "detable")):
        row = tb[k]
        setattr(s, name, _p(row, C.c_double))
    s.inum = len(sys.ilist)
    s.ilist = _p(arr(sys.ilist, np.int32), C.c_int)
    s.numneigh = _p(arr(sys.numneigh, np.int32), C.c_int)
    s.firstneigh = _p(arr(sys.firstneigh, np.int64), C.c_longlong)
    s.neigh = _p(arr(sys.neigh, np.int32), C.c_int)
    for k in ("iterations_max", "damping_type", "zodid", "fixed_iteration", "polar_gs", "polar_gs_ranked",
              "use_previous", "debug", "polar_damp", "polar_precision", "polar_gamma", "dd_cutoff"):
        setattr(s, k, getattr(st, k))
    return s, keep


def compute(sys, eflag=1, vflag=2, mu0=None, settings=None, trace=False):
    """Run the oracle's compute() on a PolarSystem; returns a dict of outputs."""
    L = lib()
    s, keep = make_struct(sys, settings)
    st = settings or sys.settings
    nall = sys.nlocal + sys.nghost
    f = np.zeros((nall, 3))
    mu = np.zeros((sys.nlocal, 3)) if mu0 is None else np.array(mu0, dtype=np.float64, copy=True)
    ef = np.zeros((sys.nlocal, 3))
    res = OrcResult()
    ut = np.zeros(st.iterations_max + 8) if trace else None
    eatom = np.zeros(nall) if eflag // 2 else None
    vatom = np.zeros((nall, 6)) if vflag // 4 else None
    rc = L.orc_compute_peratom(C.byref(s), eflag, vflag, _p(f, C.c_double), _p(mu, C.c_double),
                               _p(ef, C.c_double), C.byref(res), _p(ut, C.c_double) if trace else None,
                               _p(eatom, C.c_double) if eatom is not None else None,
                               _p(vatom, C.c_double) if vatom is not None else None)
    out = dict(f=f, mu=mu, ef_static=ef, status=rc, utrace=ut, eatom=eatom, vatom=vatom)
    for name, _ in OrcResult._fields_:
        v = getattr(res, name)
        out[name] = np.array(list(v)) if name in ("virial", "force_atom0", "dipole_force_atom0") else v
    return out


def fold_ghost_forces(f, owner, nlocal):
    """What LAMMPS' reverse_comm does after pair->compute (reference src/verlet.cpp:335):
    add ghost forces onto their owners."""
    out = np.zeros((nlocal,) + f.shape[1:])
    np.add.at(out, owner, f)
    return out
