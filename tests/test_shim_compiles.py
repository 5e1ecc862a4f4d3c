"""The LAMMPS-side files (the Pair shim and the atom style that carries the polarization attributes) must compile to
objects against the reference's headers as they are, reference nothing outside LAMMPS' core and the C-ABI, and link
against libpolar_mi355x.so.  Build container only: /root/reference does not exist on the GPU box, where this skips."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src"
PKG = os.path.join(ROOT, "lammps-induced-dipole-polarization-pair-style_amd")
SHIM = [os.path.join(ROOT, "lammps_shim", f) for f in
        ("pair_lj_cut_coul_long_polarization_mi355x.cpp", "atom_vec_full_polar.cpp")]
INC = [f"-I{REF}", f"-I{REF}/MOLECULE", f"-I{REF}/STUBS", f"-I{ROOT}/include"]

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference headers not present")

# what a LAMMPS executable provides besides its own classes: MPI (or src/STUBS) and the C / C++ runtime libraries
def _runtime_exports():
    syms = set()
    r = subprocess.run(["g++", "-print-file-name=libstdc++.so.6"], capture_output=True, text=True)
    libs = [r.stdout.strip()]
    for name in ("libc.so.6", "libm.so.6", "libgcc_s.so.1"):
        q = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
        libs.append(q)
    for lib in libs:
        if os.path.isabs(lib) and os.path.exists(lib):
            out = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True).stdout
            syms |= {ln.split()[-1].split("@")[0] for ln in out.splitlines() if ln.strip()}
    return syms


@pytest.fixture(scope="module")
def objects(tmp_path_factory):
    d = tmp_path_factory.mktemp("shim")
    objs = []
    for src in SHIM:
        o = str(d / (os.path.basename(src)[:-4] + ".o"))
        r = subprocess.run(["g++", "-c", "-fPIC", "-O2", "-std=c++11", "-Wall"] + INC + [src, "-o", o],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        objs.append(o)
    return d, objs


def _undefined(path):
    """(mangled, demangled) names of the undefined symbols of an object file"""
    raw = subprocess.run(["nm", "-u", path], capture_output=True, text=True, check=True).stdout
    names = sorted({ln.split()[-1] for ln in raw.splitlines() if ln.strip()})
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
    return list(zip(names, dem))


def test_undefined_symbols_are_lammps_core_or_the_c_abi(objects):
    _, objs = objects
    lib = os.path.join(PKG, "libpolar_mi355x.so")
    if not os.path.exists(lib):
        pytest.skip("libpolar_mi355x.so not built")
    exported = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in exported.splitlines() if ln.strip()}
    header = open(os.path.join(ROOT, "include", "polar_mi355x.h")).read()
    used_abi = set()
    runtime = _runtime_exports()
    for o in objs:
        for sym, dem in _undefined(o):
            if sym.startswith("polar_"):
                assert sym in exported, f"{sym} is not exported by libpolar_mi355x.so"
                assert re.search(r"\b" + sym + r"\s*\(", header), f"{sym} is not declared in include/polar_mi355x.h"
                used_abi.add(sym)
            else:
                ok = "LAMMPS_NS::" in dem or sym.startswith("MPI_") or sym in runtime or sym == "_GLOBAL_OFFSET_TABLE_"
                assert ok, f"{os.path.basename(o)} needs {dem!r}: neither LAMMPS core, MPI, the C/C++ runtime nor the C-ABI"
    # the pair shim really goes through the boundary for the hot path
    assert {"polar_create", "polar_set_atoms", "polar_set_neighbors", "polar_compute", "polar_pair_settings",
            "polar_set_newton", "polar_step_begin"} <= used_abi


def test_shim_links_against_the_library(objects):
    d, objs = objects
    lib = os.path.join(PKG, "libpolar_mi355x.so")
    if not os.path.exists(lib):
        pytest.skip("libpolar_mi355x.so not built")
    out = str(d / "libshim_check.so")
    # a shared object may leave LAMMPS' own symbols open; every polar_* reference must resolve against the library
    r = subprocess.run(["g++", "-shared", "-o", out] + objs + [f"-L{PKG}", "-lpolar_mi355x", f"-Wl,-rpath,{PKG}"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    needed = subprocess.run(["readelf", "-d", out], capture_output=True, text=True, check=True).stdout
    assert "libpolar_mi355x.so" in needed
