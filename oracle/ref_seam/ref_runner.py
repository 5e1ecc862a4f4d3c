"""ctypes binding of oracle/_ref/libref_seam.so -- the reference's own pair style compiled from
/root/reference (see oracle/Makefile target `ref`).  TEST INFRASTRUCTURE; build container only."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(os.path.dirname(_HERE), "_ref", "libref_seam.so")
dp, ip, llp = C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_longlong)
cpp = C.POINTER(C.c_char_p)


class SeamInput(C.Structure):
    _fields_ = [("nlocal", C.c_int), ("nghost", C.c_int), ("ntypes", C.c_int),
                ("x", dp), ("q", dp), ("alpha", dp), ("type", ip), ("molecule", ip),
                ("boxlo", C.c_double * 3), ("prd", C.c_double * 3),
                ("g_ewald", C.c_double), ("qqrd2e", C.c_double),
                ("special_lj", C.c_double * 4), ("special_coul", C.c_double * 4),
                ("inum", C.c_int), ("ilist", ip), ("numneigh", ip), ("firstneigh", llp), ("neigh", ip),
                ("nstyle", C.c_int), ("style_args", cpp), ("ncoeff", C.c_int), ("coeff_rows", cpp),
                ("nmodify", C.c_int), ("modify_args", cpp),
                ("eflag", C.c_int), ("vflag", C.c_int), ("ncalls", C.c_int), ("newton_off", C.c_int)]


class SeamOutput(C.Structure):
    _fields_ = [("f", dp), ("mu", dp), ("ef_static", dp),
                ("eng_vdwl", C.c_double), ("eng_coul", C.c_double), ("eng_pol", C.c_double),
                ("virial", C.c_double * 6), ("warnings", C.c_int), ("message", C.c_char * 256),
                ("eatom", dp), ("vatom", dp)]


def available():
    return os.path.exists("/root/reference/src/pair_lj_cut_coul_long_polarization.cpp")


def build():
    src = os.path.join(_HERE, "seam_harness.cpp")
    if not os.path.exists(_SO) or os.path.getmtime(src) > os.path.getmtime(_SO):
        subprocess.check_call(["make", "-C", os.path.dirname(_HERE), "ref"], stdout=subprocess.DEVNULL)
    return _SO


def run(sysm, style_args, coeff_rows, modify_args=(), eflag=1, vflag=2, ncalls=1, mu0=None):
    """Run the reference's compute() on a PolarSystem (workload.py).  coeff_rows: list of str."""
    lib = C.CDLL(build())
    lib.seam_run.restype = C.c_int
    keep = []

    def P(a, dt, ct):
        b = np.ascontiguousarray(a, dtype=dt)
        keep.append(b)
        return b.ctypes.data_as(C.POINTER(ct))

    def S(strs):
        arr = (C.c_char_p * max(len(strs), 1))(*[s.encode() for s in strs])
        keep.append(arr)
        return arr

    si = SeamInput()
    si.nlocal, si.nghost, si.ntypes = sysm.nlocal, sysm.nghost, sysm.ntypes
    si.x, si.q, si.alpha = P(sysm.x, np.float64, C.c_double), P(sysm.q, np.float64, C.c_double), P(sysm.alpha, np.float64, C.c_double)
    si.type, si.molecule = P(sysm.type, np.int32, C.c_int), P(sysm.molecule, np.int32, C.c_int)
    si.boxlo[:] = list(sysm.boxlo); si.prd[:] = list(sysm.prd)
    si.g_ewald, si.qqrd2e = sysm.g_ewald, sysm.qqrd2e
    si.special_lj[:] = list(sysm.special_lj); si.special_coul[:] = list(sysm.special_coul)
    si.inum = len(sysm.ilist)
    si.ilist, si.numneigh = P(sysm.ilist, np.int32, C.c_int), P(sysm.numneigh, np.int32, C.c_int)
    si.firstneigh, si.neigh = P(sysm.firstneigh, np.int64, C.c_longlong), P(sysm.neigh, np.int32, C.c_int)
    si.nstyle, si.style_args = len(style_args), S(list(style_args))
    si.ncoeff, si.coeff_rows = len(coeff_rows), S(list(coeff_rows))
    si.nmodify, si.modify_args = len(modify_args), S(list(modify_args))
    si.eflag, si.vflag, si.ncalls = eflag, vflag, ncalls
    si.newton_off = 0 if int(sysm.extra.get("newton_pair", 1)) else 1
    nall = sysm.nlocal + sysm.nghost
    f = np.zeros((nall, 3)); mu = np.zeros((sysm.nlocal, 3)) if mu0 is None else np.array(mu0, dtype=np.float64)
    ef = np.zeros((sysm.nlocal, 3))
    so = SeamOutput()
    so.f, so.mu, so.ef_static = f.ctypes.data_as(dp), mu.ctypes.data_as(dp), ef.ctypes.data_as(dp)
    eatom = np.zeros(nall) if eflag // 2 else None
    vatom = np.zeros((nall, 6)) if vflag // 4 else None
    if eatom is not None: so.eatom = eatom.ctypes.data_as(dp)
    if vatom is not None: so.vatom = vatom.ctypes.data_as(dp)
    rc = lib.seam_run(C.byref(si), C.byref(so))
    return dict(eatom=eatom, vatom=vatom, rc=rc, f=f, mu=mu, ef_static=ef, eng_vdwl=so.eng_vdwl, eng_coul=so.eng_coul, eng_pol=so.eng_pol,
                virial=np.array(list(so.virial)), warnings=so.warnings, message=so.message.decode())
