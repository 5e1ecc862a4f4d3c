"""Register / scratch budget of the hot kernels, read from hipcc's resource remarks (cross-compiles for gfx950 without a
GPU).  A kernel that starts spilling to scratch or drops below its wave occupancy loses tens of percent silently --
e.g. an early return inside the minimum-image helper once cost 24 bytes of scratch per lane in four kernels and 20 %
of the step time."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "lammps-induced-dipole-polarization-pair-style_amd", "csrc")
SRCS = [os.path.join(CSRC, f) for f in ("polar_step.hip", "polar_color.hip", "polar_api.hip", "polar_dist.hip")]   # the translation units of the library
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.fixture(scope="module")
def usage(tmp_path_factory):
    if not (os.path.exists(HIPCC) or shutil.which("hipcc")):
        pytest.skip("hipcc not available")
    out = str(tmp_path_factory.mktemp("res") / "lib.so")
    from concurrent.futures import ThreadPoolExecutor

    def one(src):
        return subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function", "-c", "-o",
                               out + os.path.basename(src) + ".o", src, "-Rpass-analysis=kernel-resource-usage"],
                              capture_output=True, text=True)

    with ThreadPoolExecutor(max_workers=4) as ex:
        runs = list(ex.map(one, SRCS))
    for r in runs:
        assert r.returncode == 0, r.stderr[-2000:]
    res, cur = {}, None
    for ln in "\n".join(r.stderr for r in runs).splitlines():
        m = re.search(r"Function Name: (\S+)", ln)
        if m:
            cur = res.setdefault(m.group(1), {})
        m = re.search(r"(VGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|VGPRs Spill): (\d+)", ln)
        if m and cur is not None:
            cur[m.group(1)] = int(m.group(2))
    assert res, "no resource remarks parsed"
    return res


def _pick(usage, *frags):
    return {k: v for k, v in usage.items() if all(f in k for f in frags)}


def test_no_kernel_uses_scratch_or_spills_vector_registers(usage):
    bad = {k: v for k, v in usage.items() if v.get("ScratchSize [bytes/lane]", 0) or v.get("VGPRs Spill", 0)}
    assert not bad, bad


def test_sweep_kernel_keeps_its_occupancy(usage):
    lp = _pick(usage, "k_field_lpI")          # the default list-mode sweep: two 4 KB LDS tiles per wave allow five
    assert lp                                 # 256-thread workgroups per CU = five waves per SIMD -> <= 96 VGPRs
    for k, v in lp.items():
        assert v["VGPRs"] <= 96, (k, v)


def test_list_and_row_kernels_stay_within_their_budgets(usage):
    for frag, cap in (("k_nl_build", 64), ("k_static_fieldILb0", 64), ("k_lj_nl_build", 64)):
        ks = _pick(usage, frag)
        assert ks, frag
        for k, v in ks.items():
            assert v["VGPRs"] <= cap, (k, v)
    for k, v in _pick(usage, "k_polar_forceILb0").items():   # FP64-heavy: four waves per SIMD at least
        assert v["VGPRs"] <= 128, (k, v)
    for k, v in _pick(usage, "k_ljcoul").items():
        assert v["VGPRs"] <= 128, (k, v)


def test_exact_mode_block_sweep_stays_light(usage):
    """k_gs_blk (exact mode: one launch per block of the sweep, csrc/polar_exact.hpp) is a stream of dot products: many waves per
    SIMD hide its one round of loads; the FP64 MFMA product keeps its accumulators in AGPRs."""
    ks = _pick(usage, "k_gs_blkILi")
    assert len(ks) == 3, list(ks)
    for k, v in ks.items():
        assert v["VGPRs"] <= 64, (k, v)
    for k, v in _pick(usage, "k_gs_gemm").items():
        assert v["VGPRs"] <= 128, (k, v)


def test_product_translation_units_never_include_lab_code():
    """the shelved kernels and host code live under csrc/lab/ and reach a translation unit only inside
    `#ifdef POLAR_LAB`.  The dependency list of every translation unit WITHOUT -DPOLAR_LAB names no file of csrc/lab/; with it,
    it does; and the built product library holds none of the lab kernels."""
    if not (os.path.exists(HIPCC) or shutil.which("hipcc")):
        pytest.skip("hipcc not available")
    seen_lab = set()
    for src in SRCS:
        for flags, want in (([], False), (["-DPOLAR_LAB"], True)):
            r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-std=c++17", "-MM", "--cuda-host-only"] + flags + [src],
                               capture_output=True, text=True, cwd=CSRC)
            assert r.returncode == 0, r.stderr[-2000:]
            deps = [d for d in r.stdout.replace("\\\n", " ").split() if "/lab/" in d or d.startswith("lab/")]
            if not want:
                assert deps == [], (os.path.basename(src), deps)
            seen_lab.update(os.path.basename(d) for d in deps)
    on_disk = {f for f in os.listdir(os.path.join(CSRC, "lab")) if f.endswith((".hpp", ".inc"))}
    assert seen_lab == on_disk, (sorted(on_disk - seen_lab), sorted(seen_lab - on_disk))    # (every lab file is reachable from the lab build, none is dead)
    so = os.path.join(os.path.dirname(CSRC), "libpolar_mi355x.so")
    if os.path.exists(so):
        blob = open(so, "rb").read()
        for name in (b"k_field_tile", b"k_field_quad", b"k_field_lpr", b"k_field_lp2", b"k_field_cl", b"k_region_sub", b"POLAR_ABLATE", b"POLAR_PIPELINE"):
            assert name not in blob, name
