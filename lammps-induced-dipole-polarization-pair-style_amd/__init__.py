"""MI355X-native ``lj/cut/coul/long/polarization`` pair style: Python host plumbing.

The product is ``libpolar_mi355x.so`` (hand-written HIP for gfx950 behind the C-ABI declared
in ``include/polar_mi355x.h``).  This module only builds / loads that library and offers
``PolarPair``, a thin mirror of the reference's ``Pair`` interface
(reference: src/pair_lj_cut_coul_long_polarization.h:30-53 -- settings, coeff, init_style/
init_one, compute, single, extract) for tests, bench.py and the multi-GPU driver.
There is no Python or CPU implementation of the hot path here: if the library or a GPU
is missing, ``PolarPair.compute`` raises.
"""
from __future__ import annotations

import ctypes as C
import importlib
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(_HERE, "libpolar_mi355x.so")
# the LAB build of the same sources (-DPOLAR_LAB): the sweep kernels that were measured and not kept as the default, the
# `ablate` timing switches and the environment knobs that select them.  tools/ and the alternative-kernel tests load it; the
# product library above contains none of that.
LIB_PATH_LAB = os.path.join(_HERE, "libpolar_mi355x_lab.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"]

_lib = None
_lib_lab = None


class PolarError(RuntimeError):
    """Raised for every negative C-ABI status; ``str(e)`` is the reference's error text."""

    def __init__(self, code, message):
        super().__init__(message)
        self.code = code


def _sources(lab=False):
    """What a library is built from: csrc/*.hip, csrc/*.hpp and the C-ABI header; the lab build also reads csrc/lab/* (kernels and
    host code that were built, measured and shelved: the product translation units never include them)."""
    src = [os.path.join(_CSRC, f) for f in sorted(os.listdir(_CSRC)) if f.endswith((".hip", ".hpp"))] + [
        os.path.join(os.path.dirname(_HERE), "include", "polar_mi355x.h")]
    labdir = os.path.join(_CSRC, "lab")
    if lab and os.path.isdir(labdir):
        src += [os.path.join(labdir, f) for f in sorted(os.listdir(labdir)) if f.endswith((".hpp", ".inc"))]
    return src


TRANSLATION_UNITS = ("polar_api.hip", "polar_step.hip", "polar_color.hip", "polar_dist.hip")


def build(force=False, verbose=False, lab=True):
    """Compile the HIP library in-tree for gfx950 (cross-compiles without a GPU): the product library and, with
    ``lab``, the lab build of the same sources.  The translation units are compiled side by side, then linked."""
    from concurrent.futures import ThreadPoolExecutor

    jobs, links = [], []
    for path, extra, tag in ((LIB_PATH, [], "product"), (LIB_PATH_LAB, ["-DPOLAR_LAB"], "lab")):
        if path == LIB_PATH_LAB and not lab:
            continue
        stale = (not os.path.exists(path)) or any(os.path.getmtime(s) > os.path.getmtime(path) for s in _sources(lab=(path == LIB_PATH_LAB)))
        if not (force or stale):
            continue
        odir = os.path.join(_CSRC, "build", tag)
        os.makedirs(odir, exist_ok=True)
        objs = []
        for tu in TRANSLATION_UNITS:
            obj = os.path.join(odir, tu.replace(".hip", ".o"))
            objs.append(obj)
            jobs.append([HIPCC] + [f for f in HIP_FLAGS if f != "-shared"] + extra + ["-Wno-unused-function", "-c", "-o", obj, os.path.join(_CSRC, tu)])
        links.append([HIPCC, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", path] + objs)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), max(1, (os.cpu_count() or 2)))) as ex:
            list(ex.map(run, jobs))
        for cmd in links:
            run(cmd)
    return LIB_PATH


class Settings(C.Structure):
    _fields_ = [("cut_lj_global", C.c_double), ("cut_coul", C.c_double), ("polar_precision", C.c_double),
                ("polar_damp", C.c_double), ("polar_gamma", C.c_double), ("iterations_max", C.c_int),
                ("damping_type", C.c_int), ("zodid", C.c_int), ("fixed_iteration", C.c_int), ("polar_gs", C.c_int),
                ("polar_gs_ranked", C.c_int), ("use_previous", C.c_int), ("debug", C.c_int), ("dd_cutoff", C.c_double),
                ("device_neigh", C.c_int), ("restart_polar", C.c_int), ("deterministic", C.c_int), ("polar_sor", C.c_double), ("rccl_halo", C.c_int), ("polar_accel", C.c_int)]


class Result(C.Structure):
    _fields_ = [("eng_vdwl", C.c_double), ("eng_coul", C.c_double), ("eng_pol", C.c_double), ("u_self", C.c_double),
                ("u_ef", C.c_double), ("u_dd", C.c_double), ("virial", C.c_double * 6), ("rmin", C.c_double),
                ("rms_dmu", C.c_double), ("iterations", C.c_int), ("sweeps", C.c_int), ("status", C.c_int),
                ("ncolors", C.c_int), ("ms_total", C.c_double), ("ms_rank", C.c_double), ("ms_ljcoul", C.c_double),
                ("ms_static", C.c_double), ("ms_solve", C.c_double), ("ms_force", C.c_double), ("ms_list", C.c_double),
                ("dd_pairs", C.c_longlong), ("ms_color_host", C.c_double)]


_dp, _ip, _llp = C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_longlong)
_cpp = C.POINTER(C.c_char_p)
EXPORTS = {
    "polar_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "polar_destroy": (C.c_int, [C.c_void_p]),
    "polar_last_error": (C.c_char_p, [C.c_void_p]),
    "polar_last_warning": (C.c_char_p, [C.c_void_p]),
    "polar_device_count": (C.c_int, []),
    "polar_kernel_version": (C.c_char_p, []),
    "polar_pair_settings": (C.c_int, [C.c_void_p, C.c_int, _cpp]),
    "polar_pair_coeff": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _cpp]),
    "polar_pair_modify": (C.c_int, [C.c_void_p, C.c_int, _cpp]),
    "polar_pair_init": (C.c_int, [C.c_void_p, C.c_double, C.c_double, _dp, _dp]),
    "polar_pair_cut": (C.c_double, [C.c_void_p, C.c_int, C.c_int]),
    "polar_pair_single": (C.c_double, [C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_int, C.c_double, C.c_double,
                                       C.c_double, _dp]),
    "polar_pair_extract": (C.c_void_p, [C.c_void_p, C.c_char_p, _ip]),
    "polar_get_settings": (C.c_int, [C.c_void_p, C.POINTER(Settings)]),
    "polar_set_settings": (C.c_int, [C.c_void_p, C.POINTER(Settings)]),
    "polar_set_types": (C.c_int, [C.c_void_p, C.c_int] + [_dp] * 7),
    "polar_set_coul": (C.c_int, [C.c_void_p, C.c_double, C.c_double, _dp, _dp, C.c_int, C.c_int, C.c_int, C.c_double]
                       + [_dp] * 8),
    "polar_set_box": (C.c_int, [C.c_void_p, _dp, _dp, _dp, _ip, C.c_int]),
    "polar_set_atoms": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _dp, _dp, _dp, _ip, _ip]),
    "polar_set_positions": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _dp]),
    "polar_set_neighbors": (C.c_int, [C.c_void_p, C.c_int, _ip, _ip, C.POINTER(_ip)]),
    "polar_set_neighbors_csr": (C.c_int, [C.c_void_p, C.c_int, _ip, _ip, _llp, _ip]),
    "polar_build_neighbors": (C.c_int, [C.c_void_p, _dp, _ip, _ip, _ip, C.c_int, _ip, C.c_int]),
    "polar_compute": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _dp, _dp, _dp, C.POINTER(Result)]),
    "polar_compute_peratom": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp, C.POINTER(Result)]),
    "polar_compute_resident": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(Result)]),
    "polar_dev_ptr": (C.c_void_p, [C.c_void_p, C.c_char_p]),
    "polar_download": (C.c_int, [C.c_void_p, C.c_char_p, _dp, C.c_longlong]),
    "polar_upload_mu": (C.c_int, [C.c_void_p, _dp, C.c_longlong]),
    "polar_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "polar_set_row_range": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "polar_set_global_count": (C.c_int, [C.c_void_p, C.c_longlong]),
    "polar_step_mu_get": (C.c_int, [C.c_void_p, C.c_longlong, C.c_longlong, _dp]),
    "polar_step_mu_put_idx": (C.c_int, [C.c_void_p, C.c_longlong, _ip, _dp]),
    "polar_step_change_get": (C.c_int, [C.c_void_p, _dp]),
    "polar_step_sweep_end_host": (C.c_int, [C.c_void_p, C.c_double]),
    "polar_step_sweep_end_n": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "polar_set_list_style": (C.c_int, [C.c_void_p, C.c_int]),
    "polar_set_newton": (C.c_int, [C.c_void_p, C.c_int]),
    "polar_get_debug_trace": (C.c_int, [C.c_void_p, _dp, C.c_int]),
    "polar_get_debug_forces": (C.c_int, [C.c_void_p, _dp]),
    "polar_get_colors": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.c_int]),
    "polar_restart_pack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "polar_restart_unpack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "polar_step_begin": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "polar_step_sweep": (C.c_int, [C.c_void_p]),
    "polar_step_sweep_part": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "polar_step_sweep_end": (C.c_int, [C.c_void_p, C.c_void_p]),
    "polar_step_state": (C.c_int, [C.c_void_p, _ip, _ip, _ip]),
    "polar_step_finish": (C.c_int, [C.c_void_p, C.POINTER(Result)]),
    "polar_mu_gather": (C.c_int, [C.c_void_p, C.c_longlong, C.c_longlong, C.c_void_p]),
    "polar_mu_scatter": (C.c_int, [C.c_void_p, C.c_longlong, C.c_longlong, C.c_void_p]),
    "polar_mu_gather_idx": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p]),
    "polar_mu_scatter_idx": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p]),
    "polar_change_export": (C.c_int, [C.c_void_p, C.c_void_p]),
    "polar_dist_unique_id": (C.c_int, [C.c_void_p]),
    "polar_dist_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "polar_dist_destroy": (C.c_int, [C.c_void_p]),
    "polar_dist_last_error": (C.c_char_p, [C.c_void_p]),
    "polar_dist_set_cadence": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "polar_dist_set_halo": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, _ip, _ip, _ip, _ip, _ip]),
    "polar_dist_set_schedule": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "polar_dist_set_ghosts": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, _ip, _dp]),
    "polar_dist_positions": (C.c_int, [C.c_void_p, C.c_void_p]),
    "polar_dist_local_result": (C.c_int, [C.c_void_p, C.POINTER(Result)]),
    "polar_dist_comm_count": (C.c_int, [C.c_void_p]),
    "polar_step_sweep_phase": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "polar_set_colors": (C.c_int, [C.c_void_p, _ip, C.c_int]),
    "polar_set_positions_range": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _dp]),
    "polar_dist_exchange": (C.c_int, [C.c_void_p, C.c_void_p]),
    "polar_dist_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(Result)]),
    "polar_dist_counters": (C.c_int, [C.c_void_p, _ip, _ip]),
    "polar_dist_profile": (C.c_int, [C.c_void_p, C.c_int]),
    "polar_dist_profile_get": (C.c_int, [C.c_void_p, _dp, _ip]),
}


def lib(lab=False):
    """Load the C-ABI library (``lab``: the lab build).  Fails loudly when it has not been built."""
    global _lib, _lib_lab
    if lab:
        if _lib_lab is None:
            if not os.path.exists(LIB_PATH_LAB):
                raise PolarError(-2, f"{LIB_PATH_LAB} is missing: run __graft_entry__.build()")
            lib()  # load order (torch first) as for the product library
            L = C.CDLL(LIB_PATH_LAB)
            for name, (res, args) in EXPORTS.items():
                fn = getattr(L, name)
                fn.restype, fn.argtypes = res, args
            _lib_lab = L
        return _lib_lab
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PolarError(-2, f"{LIB_PATH} is missing: run __graft_entry__.build() (hipcc, gfx950); "
                                 "there is no CPU fallback for the polarization hot path")
        # Load order matters: torch ships its own ROCm runtime libraries.  If this library pulled in
        # the system libamdhip64 first, a later `import torch` would bind to it and find no GPU; with
        # torch first, both share torch's runtime (same SONAME), so torch streams / tensors can be
        # handed to the kernels (polar_set_stream, parallel.py).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in EXPORTS.items():
            fn = getattr(L, name)  # AttributeError if a declared symbol is not exported
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def device_count():
    return lib().polar_device_count()


def kernel_version():
    return lib().polar_kernel_version().decode()


def _dptr(a):
    return a.ctypes.data_as(_dp)


def _iptr(a):
    return a.ctypes.data_as(_ip)


def _argv(args):
    arr = (C.c_char_p * max(len(args), 1))(*[str(a).encode() for a in args])
    return arr


class PolarPair:
    """Mirror of the reference Pair interface on top of the C-ABI (one handle = one Pair instance)."""

    def __init__(self, device=0, lab=False):
        self.L = lib(lab)
        self.h = C.c_void_p()
        rc = self.L.polar_create(device, C.byref(self.h))
        if rc < 0:
            raise PolarError(rc, "polar_create failed")
        self._keep = []
        self.nlocal = self.nghost = 0

    def close(self):
        if self.h:
            self.L.polar_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc < 0:
            raise PolarError(rc, self.L.polar_last_error(self.h).decode())
        return rc

    # ---- text interface (reference grammar) ----
    def settings(self, args):
        """pair_style arguments after the style name (reference PS.cpp:678-766)."""
        self._ck(self.L.polar_pair_settings(self.h, len(args), _argv(args)))

    def coeff(self, ntypes, args):
        """pair_coeff I J eps sigma [cut] (reference PS.cpp:772-800)."""
        self._ck(self.L.polar_pair_coeff(self.h, ntypes, len(args), _argv(args)))

    def modify(self, args):
        self._ck(self.L.polar_pair_modify(self.h, len(args), _argv(args)))

    def init(self, g_ewald, qqrd2e, special_lj=(1.0, 0.0, 0.0, 0.0), special_coul=(1.0, 0.0, 0.0, 0.0)):
        slj = np.asarray(special_lj, dtype=np.float64)
        sc = np.asarray(special_coul, dtype=np.float64)
        self._ck(self.L.polar_pair_init(self.h, g_ewald, qqrd2e, _dptr(slj), _dptr(sc)))

    def set_coul(self, g_ewald, qqrd2e, coul, special_lj=(1.0, 0.0, 0.0, 0.0), special_coul=(1.0, 0.0, 0.0, 0.0)):
        """Hand over the Coulomb lookup tables (what a LAMMPS shim takes from Pair::init_tables; tests take them from
        workload.init_coul_tables): ``coul`` = dict(nbits, mask, shift, tabinnersq, tables[8][2**nbits])."""
        slj = np.asarray(special_lj, dtype=np.float64)
        sc = np.asarray(special_coul, dtype=np.float64)
        tb = np.ascontiguousarray(coul["tables"], dtype=np.float64)
        rows = [np.ascontiguousarray(tb[k]) for k in range(8)]
        self._keep = rows
        self._ck(self.L.polar_set_coul(self.h, g_ewald, qqrd2e, _dptr(slj), _dptr(sc), int(coul["nbits"]), int(coul["mask"]),
                                       int(coul["shift"]), float(coul["tabinnersq"]), *[_dptr(r) for r in rows]))

    def restart_pack(self):
        """The polarization keywords as the record write_restart_settings appends with ``restart_polar yes``."""
        n = self.L.polar_restart_pack(self.h, None, 0)
        buf = C.create_string_buffer(n)
        self._ck(self.L.polar_restart_pack(self.h, buf, n))
        return buf.raw

    def restart_unpack(self, data):
        self._ck(self.L.polar_restart_unpack(self.h, C.c_char_p(bytes(data)), len(data)))

    def get_settings(self):
        s = Settings()
        self._ck(self.L.polar_get_settings(self.h, C.byref(s)))
        return s

    def cut(self, i, j):
        return self.L.polar_pair_cut(self.h, i, j)

    def single(self, qi, qj, itype, jtype, rsq, factor_coul=1.0, factor_lj=1.0):
        ff = C.c_double()
        e = self.L.polar_pair_single(self.h, qi, qj, itype, jtype, rsq, factor_coul, factor_lj, C.byref(ff))
        if e != e:  # NaN: the library could not evaluate the pair (message in polar_last_error)
            raise PolarError(-1, self.L.polar_last_error(self.h).decode())
        return e, ff.value

    def extract(self, name):
        dim = C.c_int()
        p = self.L.polar_pair_extract(self.h, name.encode(), C.byref(dim))
        if not p:
            return None
        if dim.value == 0:
            return C.cast(p, _dp)[0]
        n = None
        return p, dim.value

    # ---- per-run data ----
    def set_box(self, boxlo, prd, periodic=(1, 1, 1), tilt=(0.0, 0.0, 0.0), triclinic=0):
        lo, pr, ti = (np.asarray(v, dtype=np.float64) for v in (boxlo, prd, tilt))
        pe = np.asarray(periodic, dtype=np.int32)
        self._ck(self.L.polar_set_box(self.h, _dptr(lo), _dptr(pr), _dptr(ti), _iptr(pe), triclinic))

    def set_atoms(self, nlocal, nghost, x, q, alpha, typ, mol):
        x = np.ascontiguousarray(x, dtype=np.float64)
        q = np.ascontiguousarray(q, dtype=np.float64)
        alpha = np.ascontiguousarray(alpha, dtype=np.float64)
        typ = np.ascontiguousarray(typ, dtype=np.int32)
        mol = np.ascontiguousarray(mol, dtype=np.int32)
        assert x.shape == (nlocal + nghost, 3) and len(q) == len(alpha) == len(typ) == len(mol) == nlocal + nghost
        self._ck(self.L.polar_set_atoms(self.h, nlocal, nghost, _dptr(x), _dptr(q), _dptr(alpha), _iptr(typ), _iptr(mol)))
        self.nlocal, self.nghost = nlocal, nghost

    def set_colors(self, color):
        """Impose the colour phases (one int per local atom, -1 = none); ``None`` withdraws them."""
        if color is None:
            self._ck(self.L.polar_set_colors(self.h, None, 0))
            return
        c = np.ascontiguousarray(color, dtype=np.int32)
        self._ck(self.L.polar_set_colors(self.h, c.ctypes.data_as(_ip), len(c)))

    def set_positions_range(self, lo, hi, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        assert x.shape == (hi - lo, 3)
        self._ck(self.L.polar_set_positions_range(self.h, lo, hi, _dptr(x)))

    def set_positions(self, x):
        """Positions only (steps between two neighbor-list builds)."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        assert x.shape == (self.nlocal + self.nghost, 3)
        self._ck(self.L.polar_set_positions(self.h, self.nlocal, self.nghost, _dptr(x)))

    def set_neighbors_csr(self, ilist, numneigh, firstneigh, neigh):
        ilist = np.ascontiguousarray(ilist, dtype=np.int32)
        numneigh = np.ascontiguousarray(numneigh, dtype=np.int32)
        firstneigh = np.ascontiguousarray(firstneigh, dtype=np.int64)
        neigh = np.ascontiguousarray(neigh, dtype=np.int32)
        assert len(numneigh) == self.nlocal == len(firstneigh)
        self._ck(self.L.polar_set_neighbors_csr(self.h, len(ilist), _iptr(ilist), _iptr(numneigh),
                                                firstneigh.ctypes.data_as(_llp), _iptr(neigh)))

    def set_neighbors_rows(self, ilist, numneigh, rows):
        """LAMMPS layout: ``firstneigh`` as an array of row pointers (paged lists)."""
        ilist = np.ascontiguousarray(ilist, dtype=np.int32)
        numneigh = np.ascontiguousarray(numneigh, dtype=np.int32)
        ptrs = (_ip * self.nlocal)()
        keep = []
        for i, r in enumerate(rows):
            r = np.ascontiguousarray(r, dtype=np.int32)
            keep.append(r)
            ptrs[i] = _iptr(r) if len(r) else None
        self._ck(self.L.polar_set_neighbors(self.h, len(ilist), _iptr(ilist), _iptr(numneigh), ptrs))

    def load_system(self, sysm, modify_args=()):
        """Feed a workload.PolarSystem through the same calls a LAMMPS shim would make."""
        st = sysm.settings
        args = [repr(st.cut_lj_global), repr(st.cut_coul),
                "precision", repr(st.polar_precision), "max_iterations", str(st.iterations_max),
                "damp", repr(st.polar_damp), "damp_type", "exponential" if st.damping_type == 0 else "none",
                "polar_gs_ranked", "no", "polar_gs", "no",
                "fixed_iteration", "yes" if st.fixed_iteration else "no", "polar_gamma", repr(st.polar_gamma),
                "debug", "yes" if st.debug else "no", "use_previous", "yes" if st.use_previous else "no"]
        if st.zodid:
            args += ["zodid", "yes"]
        elif st.polar_gs:
            args += ["polar_gs", "yes"]
        elif st.polar_gs_ranked:
            args += ["polar_gs_ranked", "yes"]
        if st.dd_cutoff > 0:
            args += ["dd_cutoff", repr(st.dd_cutoff)]
        if getattr(st, "deterministic", 0):
            args += ["deterministic", "yes" if st.deterministic == 1 else "no"]
        if getattr(st, "polar_sor", 1.0) != 1.0:
            args += ["polar_sor", repr(float(st.polar_sor))]
        if getattr(st, "polar_accel", 0):
            args += ["polar_accel", str(int(st.polar_accel))]
        self.settings(args)
        if modify_args:
            self.modify(list(modify_args))
        return args

    def set_system(self, sysm):
        self._ck(self.L.polar_set_newton(self.h, int(sysm.extra.get("newton_pair", 1))))
        self.set_box(sysm.boxlo, sysm.prd, tilt=getattr(sysm, "tilt", (0.0, 0.0, 0.0)),
                     triclinic=int(getattr(sysm, "triclinic", 0)))
        self.set_atoms(sysm.nlocal, sysm.nghost, sysm.x, sysm.q, sysm.alpha, sysm.type, sysm.molecule)
        self.set_neighbors_csr(sysm.ilist, sysm.numneigh, sysm.firstneigh, sysm.neigh)

    def build_neighbors(self, cutneighsq, tag=None, nspecial=None, special=None, special_flag=(1, 2, 2, 2),
                        exclude_molecule_intra=False):
        """Device-side neighbor build for the LJ + Ewald-real loop (instead of set_neighbors_*)."""
        cn = np.ascontiguousarray(cutneighsq, dtype=np.float64)
        keep = [cn]
        ip = lambda a: None if a is None else (keep.append(np.ascontiguousarray(a, dtype=np.int32)) or
                                               keep[-1].ctypes.data_as(_ip))
        sf = np.ascontiguousarray(special_flag, dtype=np.int32)
        maxspecial = 0 if special is None else int(np.asarray(special).shape[1])
        self._ck(self.L.polar_build_neighbors(self.h, _dptr(cn), ip(tag), ip(nspecial), ip(special), maxspecial,
                                              sf.ctypes.data_as(_ip), int(bool(exclude_molecule_intra))))

    def build_neighbors_from_system(self, sysm, skin=None):
        """The same list make_system built on the host (same cutoff, exclusions, special bonds)."""
        wlm = importlib.import_module(__name__ + ".workload")
        w = sysm.ntypes + 1
        cutneigh = sysm.extra["cutneigh"] if skin is None else float(np.sqrt(sysm.tables["cutsq"][1:, 1:].max())) + skin
        cn = np.full((w, w), cutneigh * cutneigh)
        nsp = sp = None
        if sysm.extra.get("special"):
            nsp, sp = wlm.lammps_special_arrays(sysm.nlocal, sysm.extra["special"])
        tag = (np.asarray(sysm.owner) + 1).astype(np.int32)
        self.build_neighbors(cn, tag, nsp, sp, wlm.neighbor_special_flag(sysm.special_lj, sysm.special_coul),
                             sysm.extra.get("exclude_intra", False))

    # ---- the hot path ----
    def compute(self, eflag=1, vflag=2, mu=None, want_ef=True):
        n, nall = self.nlocal, self.nlocal + self.nghost
        f = np.zeros((nall, 3))
        mu = np.zeros((n, 3)) if mu is None else np.array(mu, dtype=np.float64, copy=True)
        ef = np.zeros((n, 3))
        res = Result()
        eatom = vatom = None
        if eflag // 2 or vflag // 4:  # per-atom tallies (Pair::eatom / Pair::vatom)
            eatom = np.zeros(nall) if eflag // 2 else None
            vatom = np.zeros((nall, 6)) if vflag // 4 else None
            rc = self._ck(self.L.polar_compute_peratom(
                self.h, eflag, vflag, _dptr(f), _dptr(mu), _dptr(ef) if want_ef else None,
                _dptr(eatom) if eatom is not None else None, _dptr(vatom) if vatom is not None else None,
                C.byref(res)))
        else:
            rc = self._ck(self.L.polar_compute(self.h, eflag, vflag, _dptr(f), _dptr(mu),
                                               _dptr(ef) if want_ef else None, C.byref(res)))
        out = _result_dict(res)
        out.update(f=f, mu=mu, ef_static=ef, status=rc, warning=self.L.polar_last_warning(self.h).decode(),
                   eatom=eatom, vatom=vatom)
        return out

    def compute_resident(self, eflag=1, vflag=2):
        res = Result()
        rc = self._ck(self.L.polar_compute_resident(self.h, eflag, vflag, C.byref(res)))
        out = _result_dict(res)
        out.update(status=rc, warning=self.L.polar_last_warning(self.h).decode())
        return out

    def debug_trace(self, nmax=1024):
        """u_polar after every sweep of the last solve (`debug yes`, reference PS.cpp:1182-1191)."""
        a = np.zeros(nmax)
        n = self._ck(self.L.polar_get_debug_trace(self.h, _dptr(a), nmax))
        return a[:n].copy()

    def debug_forces(self):
        """(polarization force on atom 0, its dipole-dipole part) of the last compute (`debug yes`, reference PS.cpp:637-638)."""
        a = np.zeros(6)
        self._ck(self.L.polar_get_debug_forces(self.h, _dptr(a)))
        return a[:3].copy(), a[3:].copy()

    def colors(self, n):
        """(ncolors, colour of every local atom in the caller's order; -1: not a row) of the last list-mode GS compute."""
        a = np.full(n, -1, dtype=np.int32)
        nc = self._ck(self.L.polar_get_colors(self.h, a.ctypes.data_as(C.POINTER(C.c_int)), n))
        return nc, a

    def download(self, name, n):
        a = np.zeros(n)
        self._ck(self.L.polar_download(self.h, name.encode(), _dptr(a), n))
        return a


class PolarDist:
    """The in-library multi-GPU driver (polar_dist_*, RCCL): one instance per rank.  ``unique_id`` = the bytes rank 0 got from
    ``PolarDist.unique_id()`` and handed to every rank (torch.distributed broadcast, MPI_Bcast, a file)."""

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        rc = lib().polar_dist_unique_id(buf)
        if rc < 0:
            raise PolarError(rc, "polar_dist_unique_id failed (is RCCL available?)")
        return buf.raw

    def __init__(self, unique_id, rank, nranks, device=0):
        self.L = lib()
        self.d = C.c_void_p()
        self.rank, self.nranks = rank, nranks
        rc = self.L.polar_dist_create(C.c_char_p(bytes(unique_id)), rank, nranks, device, C.byref(self.d))
        if rc < 0:
            raise PolarError(rc, self.L.polar_dist_last_error(self.d).decode())

    def _ck(self, rc):
        if rc < 0:
            raise PolarError(rc, self.L.polar_dist_last_error(self.d).decode())
        return rc

    def set_cadence(self, reduce_every=1, check_every=4):
        self._ck(self.L.polar_dist_set_cadence(self.d, reduce_every, check_every))

    def set_halo(self, pair, peers, send_lists, recv_lists):
        """peers[k] = rank; send_lists[k] / recv_lists[k] = handle-local atom indices (int arrays) of ``pair``'s handle."""
        peers = np.ascontiguousarray(peers, dtype=np.int32)
        sc = np.array([len(a) for a in send_lists], dtype=np.int32)
        rc_ = np.array([len(a) for a in recv_lists], dtype=np.int32)
        si = np.ascontiguousarray(np.concatenate([np.asarray(a, dtype=np.int32) for a in send_lists]) if len(send_lists) else np.zeros(0, np.int32))
        ri = np.ascontiguousarray(np.concatenate([np.asarray(a, dtype=np.int32) for a in recv_lists]) if len(recv_lists) else np.zeros(0, np.int32))
        self._ck(self.L.polar_dist_set_halo(self.d, pair.h, len(peers), _iptr(peers), _iptr(sc), _iptr(si) if len(si) else None,
                                            _iptr(rc_), _iptr(ri) if len(ri) else None))

    def set_schedule(self, lag=1, my_class=0, nclasses=0):
        """``nclasses`` > 0: one colouring shared by the ranks (this rank colours in turn ``my_class``), per-phase exchanges
        ``lag`` phases late at most; ``nclasses`` 0 or ``lag`` -1: one exchange per sweep (block-Jacobi across ranks)."""
        self._ck(self.L.polar_dist_set_schedule(self.d, lag, my_class, nclasses))

    def set_ghosts(self, pair, owner, shift):
        owner = np.ascontiguousarray(owner, dtype=np.int32)
        shift = np.ascontiguousarray(shift, dtype=np.float64)
        self._ck(self.L.polar_dist_set_ghosts(self.d, pair.h, len(owner), _iptr(owner) if len(owner) else None, _dptr(shift) if len(owner) else None))

    def positions(self, pair):
        self._ck(self.L.polar_dist_positions(self.d, pair.h))

    def local_result(self):
        res = Result()
        self._ck(self.L.polar_dist_local_result(self.d, C.byref(res)))
        return _result_dict(res)

    def comm_count(self):
        return self._ck(self.L.polar_dist_comm_count(self.d))

    PROFILE_PARTS = ("other", "sweep_kernels", "stop_rule", "exchange", "accel")

    def profile(self, enable=True):
        """Timed events between the parts of the sweep loop of the following steps (a few us each: not inside a timed region)."""
        self._ck(self.L.polar_dist_profile(self.d, 1 if enable else 0))

    def profile_get(self):
        """Device time of the last profiled step's sweep loop by part, in ms: {other (waits for exchanges on the communication
        stream, host looks at the loop state), sweep_kernels, stop_rule, exchange (on the compute stream), accel}, + intervals."""
        ms = (C.c_double * len(self.PROFILE_PARTS))()
        n = C.c_int()
        self._ck(self.L.polar_dist_profile_get(self.d, ms, C.byref(n)))
        out = {k: float(v) for k, v in zip(self.PROFILE_PARTS, ms)}
        out["intervals"] = n.value
        return out

    def exchange(self, pair):
        self._ck(self.L.polar_dist_exchange(self.d, pair.h))

    def step(self, pair, eflag=1, vflag=2):
        res = Result()
        rc = self._ck(self.L.polar_dist_step(self.d, pair.h, eflag, vflag, C.byref(res)))
        out = _result_dict(res)
        ex, ar = C.c_int(), C.c_int()
        self.L.polar_dist_counters(self.d, C.byref(ex), C.byref(ar))
        out.update(status=rc, warning=pair.L.polar_last_warning(pair.h).decode(), exchanges=ex.value, allreduces=ar.value)
        return out

    def close(self):
        if self.d:
            self.L.polar_dist_destroy(self.d)
            self.d = C.c_void_p()


def _result_dict(res):
    out = {}
    for name, _ in Result._fields_:
        v = getattr(res, name)
        out[name] = np.array(list(v)) if name == "virial" else v
    return out


def pair_from_system(sysm, coeff_rows=None, modify_args=(), device=0, device_neigh=False, row_range=None, lab=False):
    """Build a PolarPair from a workload.PolarSystem the way an input script would:
    pair_style -> pair_modify -> pair_coeff -> init -> per-step data.
    ``device_neigh``: the LJ/Coulomb list is built on the device (keyword ``device_neigh yes`` of the shim)
    instead of being uploaded; systems made with ``build_list=False`` need it.
    ``row_range`` = (lo, hi): a sharded handle (polar_set_row_range) -- set before the device list is built, which
    then covers the own rows only."""
    p = PolarPair(device, lab=lab)
    p.load_system(sysm, modify_args)
    rows = coeff_rows if coeff_rows is not None else sysm.extra.get("coeff_rows")
    if rows is None:
        raise ValueError("pair_coeff rows are required")
    for r in rows:
        p.coeff(sysm.ntypes, list(r))
    p.init(sysm.g_ewald, sysm.qqrd2e, sysm.special_lj, sysm.special_coul)
    if sysm.coul["nbits"]:  # the role of Pair::init_tables (PS.cpp:851) is played by workload.init_coul_tables
        p.set_coul(sysm.g_ewald, sysm.qqrd2e, sysm.coul, sysm.special_lj, sysm.special_coul)
    if row_range is not None:
        p._ck(p.L.polar_set_row_range(p.h, int(row_range[0]), int(row_range[1])))
    if device_neigh:
        p._ck(p.L.polar_set_newton(p.h, int(sysm.extra.get("newton_pair", 1))))
        p.set_box(sysm.boxlo, sysm.prd, tilt=getattr(sysm, "tilt", (0.0, 0.0, 0.0)), triclinic=int(getattr(sysm, "triclinic", 0)))
        p.set_atoms(sysm.nlocal, sysm.nghost, sysm.x, sysm.q, sysm.alpha, sysm.type, sysm.molecule)
        p.build_neighbors_from_system(sysm)
    else:
        p.set_system(sysm)
    return p
