import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


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
