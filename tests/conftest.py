import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """The HIP library is built in-tree by __graft_entry__.build(); a fresh checkout that runs the tests first gets it
    built here (hipcc cross-compiles gfx950 without a GPU), so that the ABI tests never pass or fail by accident."""
    so = os.path.join(ROOT, "harkdb_amd", "libhark.so")
    if not os.path.exists(so):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "harkdb_amd", "csrc")], stdout=subprocess.DEVNULL)
    yield


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def data_csv():
    """The reference's 7x8 fixture as an int64 matrix (header skipped)."""
    return np.loadtxt(os.path.join(GOLDEN, "data.csv"), delimiter=",", skiprows=1, dtype=np.int64)


def resolve_table(spec):
    if isinstance(spec, str):
        assert spec == "data.csv"
        return data_csv()
    a = np.asarray(spec, dtype=np.int64)
    return a.reshape(0, 0) if a.size == 0 else a


def expand_runs(runs, width):
    rows = []
    for cnt, row in runs:
        rows += [row] * cnt
    return np.asarray(rows, dtype=np.int64).reshape(-1, width)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as ora
    ora.lib()
    return ora
