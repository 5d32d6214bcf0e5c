"""A 32-bit key column that is in order already (a table kept in key order, a dimension table sorted by its primary key, the result
of a GROUP BY): the first read of the sort (multi_hist_kernel, k_sort.hip) notices it and no radix pass runs -- the stable sort of
a sorted sequence is the sequence.  ONE descent anywhere must be seen: positions at lane, wave, workgroup and slice boundaries."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from harkdb_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def _check(eng, keys, descending=False):
    n = len(keys)
    pay = np.arange(n, dtype=np.int32) * 3 + 1
    t = eng.table_from_columns([keys, pay])
    res = eng.sort(t, 0, [0, 1], descending=descending)
    k = keys.view(np.uint32).astype(np.int64) if keys.dtype == np.uint32 else keys.astype(np.int64)
    order = np.argsort(-k if descending else k, kind="stable")
    assert np.array_equal(res.column(0), keys[order]) and np.array_equal(res.column(1), pay[order])
    res.free(); t.free()


@pytest.mark.parametrize("n", [5, 4096, 4099, 262_147, 1_000_003, 5_000_001])
@pytest.mark.parametrize("dt", [np.uint32, np.int32])
def test_sorted_columns_with_ties_and_one_descent_at_every_kind_of_boundary(eng, n, dt):
    rng = np.random.default_rng(n)
    info = np.iinfo(dt)
    base = np.sort(rng.integers(info.min, info.max, size=n, dtype=np.int64).astype(dt))
    base[n // 3: n // 3 + min(50, n // 4)] = base[n // 3]                          # a run of equal keys: their payloads keep their order
    _check(eng, base)
    _check(eng, base[::-1].copy(), descending=True)
    _check(eng, base[::-1].copy())                                               # descending input, ascending sort: every pair is a descent
    spots = sorted({0, 1, 2, 3, 4, 62, 63, 64, 255, 256, 4094, 4095, 4096, n // 2, n - 3, n - 2} & set(range(n - 1)))
    spots += [int(x) for x in rng.integers(0, n - 1, size=6)]
    for at in spots:                                                             # keys[at] > keys[at + 1]: the only descent of the column
        k = base.copy()
        lo, hi = int(k[at]), int(k[at + 1])
        if lo == info.max: continue
        k[at] = hi + 1 if hi < info.max else hi
        if int(k[at]) <= int(k[at + 1]):
            k[at + 1] = lo - 1 if lo > info.min else lo
        if int(k[at]) <= int(k[at + 1]): continue
        _check(eng, k)


def test_join_on_a_build_side_sorted_by_its_primary_key(eng):
    import test_gpu_hjoin as T
    rng = np.random.default_rng(5)
    rk = np.sort(rng.choice(1 << 30, size=60_000, replace=False).astype(np.uint32))
    lk = rng.integers(0, 1 << 30, size=400_000).astype(np.uint32)
    hit = rng.random(len(lk)) < 0.5
    lk[hit] = rk[rng.integers(0, len(rk), size=int(hit.sum()))]
    assert T._check(eng, lk, rk) > 0
    assert T._check(eng, lk, rk[::-1].copy()) > 0


def test_the_knob_runs_the_passes_anyway():
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import numpy as np, test_gpu_presorted as P\n"
            "from harkdb_amd.engine import Engine\neng = Engine(0)\nP._check(eng, np.arange(300_000, dtype=np.uint32) // 3)\nprint('ok')\n") % (ROOT, os.path.join(ROOT, "tests"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, HARK_SORT_NO_PRESORTED="1"))
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout + out.stderr
