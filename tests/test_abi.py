"""The C-ABI library loads and exports every symbol include/hark.h declares
(no compute calls: there is no GPU in the build container)."""
import os
import re

import pytest

from conftest import ROOT


def declared_symbols(header="hark.h", prefix="hark_"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(" + prefix + r"[a-z0-9_]+)\s*\(", text)))


def test_futhark_named_api_is_declared_bound_and_exported():
    """include/futhark_compat.h = the names `futhark c --library futhark/main.fut` generates (setup.sh:12), which
    futhark_ffi.Futhark(_main) binds (FutharkContext.py:31-41): every one is exported by libhark.so."""
    from harkdb_amd import _ffi
    names = declared_symbols("futhark_compat.h", "futhark_")
    assert set(names) == set(_ffi.FUTHARK_SIGNATURES)
    for need in ("futhark_context_new", "futhark_new_i32_2d", "futhark_new_u32_2d", "futhark_values_i32_2d", "futhark_values_u32_2d",
                 "futhark_shape_i32_2d", "futhark_free_u32_2d", "futhark_entry_query_sel", "futhark_entry_query_groupby"):
        assert need in names
    lib = _ffi.bind_futhark_names(_ffi.load())
    for name in names:
        assert getattr(lib, name) is not None


def test_c_caller_compiles_against_the_headers(tmp_path):
    """tests/abi_smoke.c (the program INTEGRATION.md shows) compiles and links with plain gcc -- no HIP or C++ in the
    headers; it is RUN by the -m gpu suite (tests/test_gpu_abi_c.py)."""
    import subprocess
    subprocess.check_call(["gcc", "-O1", "-Wall", "-Werror", "-std=c99", os.path.join(ROOT, "tests", "abi_smoke.c"), "-I", os.path.join(ROOT, "include"),
                           "-L", os.path.join(ROOT, "harkdb_amd"), "-lhark", "-Wl,-rpath," + os.path.join(ROOT, "harkdb_amd"),
                           "-o", str(tmp_path / "abi_smoke")])


def test_header_symbols_all_bound():
    from harkdb_amd import _ffi
    assert set(declared_symbols()) == set(_ffi.SIGNATURES)


def test_library_exports_every_symbol():
    from harkdb_amd import _ffi
    lib = _ffi.load()
    for name in declared_symbols():
        assert getattr(lib, name) is not None
    assert lib.hark_version() >= 100


def test_no_gpu_fails_loudly():
    """Without a HIP device the product path raises instead of falling back."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from harkdb_amd.engine import Engine
    from harkdb_amd._ffi import HarkError
    with pytest.raises(HarkError):
        Engine(0)


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "harkdb_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("no CPU fallback", ""), f"{f} mentions the oracle"


def test_library_is_the_build_of_the_sources_in_the_tree():
    """The Makefile records the digest of the sources it linked (harkdb_amd/libhark.srchash); a library left behind by a build
    of other sources -- an experiment, an older checkout -- is not what the tests and the bench should run."""
    import subprocess
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "harkdb_amd", "csrc")], stdout=subprocess.DEVNULL)
    from harkdb_amd._srchash import library_matches_sources
    assert library_matches_sources() is True


@pytest.mark.gpu
def test_the_library_on_the_gpu_box_is_the_build_of_its_sources():
    """The same on the GPU box, WITHOUT building there: the snapshot brought the library, its record and the sources together."""
    from harkdb_amd._srchash import library_matches_sources
    assert library_matches_sources() is True


def test_build_checks_the_hand_placed_waits_behind_inline_asm_loads(tmp_path):
    """harkdb_amd/csrc/Makefile runs tools/check_hidden_loads.py over the units that issue loads from inline assembly (k_hjoin.hip's
    partition kernel): the product passes, and the same unit with its hand-placed waits removed is refused."""
    import shutil
    import subprocess
    import sys
    from conftest import ROOT
    tool = os.path.join(ROOT, "tools", "check_hidden_loads.py")
    good = subprocess.run([sys.executable, tool, os.path.join(ROOT, "harkdb_amd", "csrc", "k_hjoin.hip")], capture_output=True, text=True, timeout=600)
    assert good.returncode == 0 and " 0 violations" in good.stdout and "72 inline-asm load sites" in good.stdout      # (the partition kernel's batch loop is compiled twice since round 6 -- with and without the hot set's probe -- and the kernel twice: with and without rotated loads), good.stdout + good.stderr
    bad_root = tmp_path / "tree"
    shutil.copytree(os.path.join(ROOT, "harkdb_amd", "csrc"), bad_root / "harkdb_amd" / "csrc", ignore=shutil.ignore_patterns("*.o"))
    shutil.copytree(os.path.join(ROOT, "include"), bad_root / "include")
    unit = bad_root / "harkdb_amd" / "csrc" / "k_hjoin.hip"
    src = unit.read_text()
    marker = "        if (!HIDDEN) return;\n        constexpr int kAheadLoads"
    assert marker in src
    unit.write_text(src.replace(marker, "        if (!HIDDEN || tid >= 0) return;\n        constexpr int kAheadLoads"))       # the waits are gone
    bad = subprocess.run([sys.executable, tool, str(unit)], capture_output=True, text=True, timeout=600)
    assert bad.returncode == 1 and "while an inline-asm load into them is in flight" in bad.stderr, bad.stdout + bad.stderr[-500:]


def test_the_bounds_checked_build_of_the_i64_sort_still_compiles(tmp_path):
    """k_msort.hip has a build for experiments in which every global and LDS index of its kernels is checked and a violation is
    RECORDED instead of faulting (-DHARK_MSD_CHECK, tools/ab_build.sh): it found the workgroup that lost half its waves
    (profiles/r05_notes.md 11).  A build nobody compiles rots; this compiles it (device code for gfx950, no GPU needed)."""
    import subprocess
    from conftest import ROOT
    src = os.path.join(ROOT, "harkdb_amd", "csrc", "k_msort.hip")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-O1", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-DHARK_MSD_CHECK", "-I", os.path.join(ROOT, "include"),
                        "-c", src, "-o", str(tmp_path / "k_msort_chk.o")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]


def test_no_second_library_beside_the_product():
    """A/B builds go to build/ab/ (tools/ab_build.sh): every file of harkdb_amd/ travels to the GPU box with each call, and
    HARK_LIB can swap any libhark*.so for the product -- five stale ones sat here in round 5 (18 MB per push)."""
    import glob
    libs = sorted(os.path.basename(f) for f in glob.glob(os.path.join(ROOT, "harkdb_amd", "libhark*.so")))
    assert libs in ([], ["libhark.so"]), libs
