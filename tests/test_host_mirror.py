"""CPU tests of the host side above the C-ABI: the library loads without a GPU, exports every
symbol include/polar_mi355x.h declares, mirrors the reference's pair_style / pair_coeff grammar
and error strings (PS.cpp:678-800), and fails loudly instead of falling back to a CPU path."""
import math
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(pkg):
    hdr = open(os.path.join(ROOT, "include", "polar_mi355x.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(polar_[a-z_]+)\s*\(", hdr))
    assert len(names) >= 25
    L = pkg.lib()
    for n in sorted(names):
        assert hasattr(L, n), n
        assert n in pkg.EXPORTS, f"{n} has no ctypes prototype"


def test_defaults_match_reference(pkg):
    p = pkg.PolarPair(0)
    p.settings(["10.0"])
    s = p.get_settings()
    # PS.cpp:65-78
    assert (s.iterations_max, s.damping_type, s.zodid, s.fixed_iteration) == (50, 1, 0, 0)
    assert (s.polar_gs, s.polar_gs_ranked, s.use_previous, s.debug) == (0, 1, 0, 0)
    assert s.polar_damp == 2.1304 and s.polar_precision == 1e-11 and s.polar_gamma == 1.03
    assert s.cut_lj_global == 10.0 and s.cut_coul == 10.0  # one cutoff -> both (PS.cpp:683)


ARG_CASES = [
    ["2.5", "12.8345", "precision", "0.00000000001", "max_iterations", "100", "damp_type", "exponential", "damp",
     "2.1304", "polar_gs_ranked", "yes", "debug", "no", "use_previous", "yes"],
    ["2.5", "6", "max_iterations", "30", "damp_type", "exponential", "polar_gs_ranked", "yes"],
    ["9.0", "9.0", "polar_gs_ranked", "no", "polar_gs", "yes", "polar_gamma", "1.0"],
    ["9.0", "9.0", "polar_gs_ranked", "no", "zodid", "yes"],
    ["9.0", "9.0", "fixed_iteration", "yes", "max_iterations", "7", "damp_type", "none"],
    ["9.0", "9.0", "dd_cutoff", "8.5"],
    ["9.0", "9.0", "dd_cutoff", "8.5", "device_neigh", "yes"],
    ["9.0", "9.0", "dd_cutoff", "8.5", "deterministic", "yes", "polar_sor", "1.15", "restart_polar", "yes", "rccl_halo", "yes"],
]


@pytest.mark.parametrize("args", ARG_CASES)
def test_settings_parser_agrees_with_python_mirror(args, pkg, wl):
    p = pkg.PolarPair(0)
    p.settings(args)
    s = p.get_settings()
    ref = wl.parse_pair_style_args(args)
    for k in ("cut_lj_global", "cut_coul", "iterations_max", "damping_type", "zodid", "fixed_iteration", "polar_gs",
              "polar_gs_ranked", "use_previous", "debug", "polar_damp", "polar_precision", "polar_gamma", "dd_cutoff",
              "device_neigh", "deterministic", "polar_sor", "restart_polar", "rccl_halo"):
        assert getattr(s, k) == getattr(ref, k), k


ERR_CASES = [
    ([], "Illegal pair_style command"),
    (["9", "9", "zodid", "yes"], "Zodid doesn't work with polar_gs or polar_gs_ranked"),  # ranked is on by default
    (["9", "9", "polar_gs", "yes"], "polar_gs and polar_gs_ranked are mutually exclusive"),
    (["9", "9", "polar_gs_ranked", "no", "polar_gs", "yes", "polar_gs_ranked", "yes"],
     "polar_gs and polar_gs_ranked are mutually exclusive"),
    (["9", "9", "precision"], "Illegal pair_style command"),
    (["9", "9", "damp_type", "thole"], "Illegal pair_style command"),
    (["9", "9", "debug", "maybe"], "Illegal pair_style command"),
    (["9", "9", "device_neigh", "1"], "Illegal pair_style command"),
    (["9", "9", "nonsense", "1"], "Illegal pair_style command"),
    (["9", "precision", "1e-8"], "Expected floating point parameter in input script or data file"),  # scan starts at arg 2
]


@pytest.mark.parametrize("args,msg", ERR_CASES)
def test_settings_errors_carry_reference_text(args, msg, pkg):
    p = pkg.PolarPair(0)
    with pytest.raises(pkg.PolarError) as e:
        p.settings(args)
    assert str(e.value) == msg and e.value.code == -1


def test_coeff_mixing_and_init_one(pkg, wl):
    p = pkg.PolarPair(0)
    p.settings(["2.5", "12.0"])
    with pytest.raises(pkg.PolarError, match="Incorrect args for pair coefficients"):
        p.coeff(3, ["1", "1", "0.1"])
    p.coeff(3, ["1", "1", "0.10", "3.0", "9.0"])
    p.coeff(3, ["2*3", "2*3", "0.20", "3.5"])      # wildcard bounds, global LJ cutoff 2.5
    p.coeff(3, ["1", "3", "0.05", "3.3", "13.0"])  # explicit cross term with cut_lj > cut_coul
    p.init(0.21, wl.QQR2E_REAL)
    p.set_coul(0.21, wl.QQR2E_REAL, wl.init_coul_tables(12.0, 0.21, wl.QQR2E_REAL))  # the tables Pair::init_tables would hand over
    # geometric mixing for the unset 1-2 pair (src/pair.cpp:660-690), cutoff = max(cut_lj, cut_coul)
    assert p.cut(1, 2) == 12.0 and p.cut(1, 3) == 13.0 and p.cut(2, 3) == 12.0
    eps12, sig12, cut12 = math.sqrt(0.1 * 0.2), math.sqrt(3.0 * 3.5), math.sqrt(9.0 * 2.5)
    rsq = 4.0 ** 2
    e, ff = p.single(0.0, 0.0, 1, 2, rsq)          # no charges: pure LJ
    r6 = (sig12 ** 2 / rsq) ** 3
    assert abs(e - 4 * eps12 * (r6 * r6 - r6)) < 1e-14
    assert abs(ff - 24 * eps12 * (2 * r6 * r6 - r6) / rsq) < 1e-14
    e2, _ = p.single(0.0, 0.0, 1, 2, (cut12 + 0.1) ** 2)
    assert e2 == 0.0
    with pytest.raises(pkg.PolarError, match="All pair coeffs are not set"):
        q = pkg.PolarPair(0)
        q.settings(["2.5", "12.0"])
        q.coeff(2, ["1", "1", "0.1", "3.0"])
        q.init(0.2, wl.QQR2E_REAL)


def test_coulomb_tables_match_reference_scheme(pkg, wl):
    """polar_pair_single through the 12-bit table vs the erfc closed form, and vs the python
    restatement of init_tables (workload.init_coul_tables)."""
    from scipy.special import erfc

    g, cut = 0.195492, 12.8345
    p = pkg.PolarPair(0)
    p.settings(["2.5", repr(cut)])
    p.coeff(1, ["1", "1", "0.0", "1.0"])
    p.init(g, wl.QQR2E_REAL)
    tab = wl.init_coul_tables(cut, g, wl.QQR2E_REAL)
    # the library does not generate the tables (Pair::init_tables stays LAMMPS host code) and says so
    with pytest.raises(pkg.PolarError, match="Coulomb tables were not handed over"):
        p.single(0.4, -0.7, 1, 1, 9.0)
    p.set_coul(g, wl.QQR2E_REAL, tab)
    for r in (1.2, 1.5, 2.0, 3.7, 7.9, 12.8):
        rsq = r * r
        e, ff = p.single(0.4, -0.7, 1, 1, rsq)
        exact = wl.QQR2E_REAL * 0.4 * -0.7 * erfc(g * r) / r
        bare = abs(wl.QQR2E_REAL * 0.4 * 0.7 / r)
        assert abs(e - exact) / bare < 2e-6          # linear-table accuracy (erfc polynomial below tabinner)
        if rsq > tab["tabinnersq"]:
            rf = np.float32(rsq)
            it = (int(np.array([rf]).view(np.int32)[0]) & tab["mask"]) >> tab["shift"]
            fr = (float(rf) - tab["tables"][0][it]) * tab["tables"][1][it]
            e_tab = 0.4 * -0.7 * (tab["tables"][6][it] + fr * tab["tables"][7][it])
            assert abs(e - e_tab) <= 1e-13 * abs(e_tab)


def test_compute_fails_loudly_without_gpu_or_setup(pkg):
    p = pkg.PolarPair(0)
    p.settings(["2.5", "9.0"])
    if pkg.device_count() == 0:
        with pytest.raises(pkg.PolarError) as e:
            p.set_atoms(1, 0, np.zeros((1, 3)), np.zeros(1), np.zeros(1), np.ones(1, np.int32), np.ones(1, np.int32))
        assert e.value.code == -2 and "no CPU fallback" in str(e.value)
    with pytest.raises(pkg.PolarError):
        p.compute_resident()


def test_triclinic_box_is_accepted_for_exact_mode(pkg):
    p = pkg.PolarPair(0)
    p.set_box([0, 0, 0], [10, 10, 10], tilt=(1.0, 0.5, -0.5), triclinic=1)   # exact-mode kernels handle it


def test_restart_record_of_the_polarization_keywords(pkg):
    """with `restart_polar yes` the keywords travel in a tagged record behind the stock restart
    fields; a reader finds the record by its magic and leaves a reference-format stream alone."""
    import struct

    p = pkg.PolarPair(0)
    p.settings(["2.5", "12.8345", "precision", "1e-9", "max_iterations", "77", "damp_type", "exponential", "damp", "1.9",
                "polar_gamma", "1.01", "use_previous", "yes", "dd_cutoff", "11.5", "restart_polar", "yes",
                "polar_gs_ranked", "no", "polar_gs", "yes", "deterministic", "yes", "polar_sor", "1.2", "rccl_halo", "yes"])
    rec = p.restart_pack()
    magic, ver, nb = struct.unpack("<iii", rec[:12])
    assert magic == 0x524C4F50 and ver == 2 and nb == len(rec) - 12   # version 2: + deterministic, rccl_halo, polar_accel, polar_sor
    q = pkg.PolarPair(0)
    q.settings(["2.5", "12.8345"])          # what read_restart_settings does first: cutoffs from the stock record
    assert q.get_settings().polar_gs_ranked == 1 and q.get_settings().restart_polar == 0
    q.restart_unpack(rec)
    a, b = p.get_settings(), q.get_settings()
    for k, _ in pkg.Settings._fields_:
        assert getattr(a, k) == getattr(b, k), k
    assert b.deterministic == 1 and b.rccl_halo == 1 and abs(b.polar_sor - 1.2) < 1e-15
    # a version-1 record (round 3's files) is still read; the keywords it does not carry keep the values in force
    v1 = struct.pack("<iii", 0x524C4F50, 1, 72) + rec[12:12 + 72]
    q1 = pkg.PolarPair(0)
    q1.settings(["2.5", "12.8345"])
    q1.restart_unpack(v1)
    c = q1.get_settings()
    assert c.iterations_max == 77 and c.polar_gs == 1 and c.dd_cutoff == 11.5 and c.deterministic == 0 and c.polar_sor == 1.0
    # polar_set_settings: a zero-initialised struct means omega = 1, values outside (0, 2) are refused
    z = pkg.Settings()
    z.cut_lj_global, z.cut_coul, z.iterations_max, z.polar_gs_ranked = 2.5, 12.8345, 50, 1
    q1._ck(q1.L.polar_set_settings(q1.h, z))
    assert q1.get_settings().polar_sor == 1.0
    z.polar_sor = 2.5
    with pytest.raises(pkg.PolarError, match="polar_sor"):
        q1._ck(q1.L.polar_set_settings(q1.h, z))
    # a stream that continues with something else (reference-format file): refused, nothing changes
    r = pkg.PolarPair(0)
    r.settings(["2.5", "12.8345"])
    with pytest.raises(pkg.PolarError, match="not a polarization restart record"):
        r.restart_unpack(struct.pack("<iii", 7, 1, 72) + bytes(72))
    with pytest.raises(pkg.PolarError, match="unknown version or length"):
        r.restart_unpack(rec[:12] + rec[12:40])
    assert r.get_settings().iterations_max == 50
    # the keyword itself follows the grammar of the other yes/no keywords
    with pytest.raises(pkg.PolarError, match="Illegal pair_style command"):
        r.settings(["2.5", "12.8345", "restart_polar", "maybe"])
