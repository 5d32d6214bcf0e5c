"""The C-ABI library loads and exports every symbol include/hark.h declares
(no compute calls: there is no GPU in the build container)."""
import os
import re

import pytest

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "hark.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hark_[a-z0-9_]+)\s*\(", text)))


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
