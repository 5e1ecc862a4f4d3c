"""The LAMMPS-side shim must compile against the reference's headers as they are (build container
only: /root/reference does not exist on the GPU box, where this test skips)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src"


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference headers not present")
def test_shim_compiles_against_reference_headers():
    src = os.path.join(ROOT, "lammps_shim", "pair_lj_cut_coul_long_polarization_mi355x.cpp")
    r = subprocess.run(["g++", "-fsyntax-only", "-std=c++11", "-Wall", f"-I{REF}", f"-I{REF}/STUBS",
                        f"-I{ROOT}/include", src], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
