"""The join of a probe column that is sorted or clustered by the join key (harkdb_amd/csrc/k_cjoin.hip: rows searched in row
order against the sorted build side) against the reference's order -- ascending key, left row, right row (join.fut:55-75) --
by the numpy model of tests/test_gpu_hjoin.py (pinned to the oracle in tests/test_gpu_groupby_join.py)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
import test_gpu_hjoin as T

pytestmark = pytest.mark.gpu

CLUSTERED = "clustered probe column, searched in row order"


@pytest.fixture(scope="module")
def eng():
    from harkdb_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def _shape(lk, shape, rng):
    if shape == "sorted": return np.sort(lk)
    if shape == "descending": return np.sort(lk)[::-1].copy()
    if shape == "noisy":                                              # sorted, then one row in twenty overwritten with a key from anywhere (late rows)
        out = np.sort(lk)
        idx = rng.integers(0, len(lk), size=len(lk) // 20)
        out[idx] = lk[idx]
        return out
    if shape == "blocks":                                             # sorted inside blocks of 65536 rows
        out = lk.copy()
        for a in range(0, len(out), 65536): out[a:a + 65536] = np.sort(out[a:a + 65536])
        return out
    if shape == "runs":                                               # sorted, then runs of 8192 rows shuffled as wholes
        s = np.sort(lk)
        k = len(s) // 8192
        head = s[:k * 8192].reshape(k, 8192)[rng.permutation(k)].reshape(-1)
        return np.concatenate([head, s[k * 8192:]])
    raise AssertionError(shape)


@pytest.mark.parametrize("shape", ["sorted", "descending", "noisy", "blocks", "runs"])
@pytest.mark.parametrize("case", ["u32 10% hits", "u32 every row hits, 8 partners each", "i64 duplicates both sides", "i64 signed keys",
                                  "u32 unique build keys (primary key): no counts, no expansion", "i32 negative keys as u32",
                                  "u32 few distinct keys (duplicate splitters)"])
def test_clustered_probe_columns_reference_order(eng, case, shape):
    lk, rk = T._make(case, 7 + len(shape))
    lk = _shape(lk, shape, np.random.default_rng(5))
    pairs = T._check(eng, lk, rk)
    assert eng.last_join_path() == CLUSTERED, eng.last_join_path()
    if T.CASES[case]["hit"] >= 0.1: assert pairs > 0


@pytest.mark.parametrize("case", ["u32 10% hits", "i64 duplicates both sides"])
def test_short_sorted_runs_in_shuffled_order(eng, case):
    """Sorted runs of 256 rows in shuffled order: rows half a batch apart have nothing in common, but a run lands in ONE bucket
    of the partition and overruns the workgroup's slab there; rows four lanes apart are neighbours in key order, which is what
    the search path needs."""
    lk, rk = T._make(case, 11)
    s = np.sort(lk)
    k = len(s) // 256
    lk = np.concatenate([s[:k * 256].reshape(k, 256)[np.random.default_rng(2).permutation(k)].reshape(-1), s[k * 256:]])
    assert T._check(eng, lk, rk) > 0
    assert eng.last_join_path() == CLUSTERED, eng.last_join_path()


def test_shuffled_rows_of_the_same_column_stay_partitioned(eng):
    lk, rk = T._make("u32 10% hits", 3)
    assert T._check(eng, np.sort(lk), rk) > 0 and eng.last_join_path() == CLUSTERED
    assert T._check(eng, lk, rk) > 0 and eng.last_join_path().startswith("partitioned")


@pytest.mark.parametrize("dt", [np.uint32, np.int64])
@pytest.mark.parametrize("shape", ["sorted", "blocks"])
def test_sorted_probe_with_carried_and_rank_ordered_columns_only(eng, dt, shape):
    """Unique build keys and only one 4-byte column of either side selected: a sorted probe column's rows ARE the output
    order, the emit kernel writes the two columns and leaves (rank, left row) out; block by block the ranks do not ascend
    and the general ordering delivers the rows."""
    rng = np.random.default_rng(41)
    n, s = 400_003, 50_000
    info = np.iinfo(dt)
    rk = np.unique(rng.integers(info.min, info.max, size=s + 64, dtype=np.int64).astype(dt))[:s]
    rk = rk[rng.permutation(len(rk))]
    lk = rng.integers(info.min, info.max, size=n, dtype=np.int64).astype(dt)
    hit = rng.random(n) < 0.5
    lk[hit] = rk[rng.integers(0, len(rk), size=int(hit.sum()))]
    lk = _shape(lk, shape, rng)
    la, ra = rng.integers(-2**31, 2**31, n).astype(np.int32), rng.integers(0, 2**31, len(rk)).astype(np.int32)
    t1, t2 = eng.table_from_columns([lk, la]), eng.table_from_columns([rk, ra])
    res = eng.join(t1, t2, 0, 0, [1], [1])
    assert eng.last_join_path() == CLUSTERED
    li, ri = T._np_join_rows(lk, rk)
    assert res.shape == (len(li), 2) and len(li) > 0
    assert np.array_equal(res.column(0), la[li]) and np.array_equal(res.column(1), ra[ri])
    res.free(); t1.free(); t2.free()


def test_sorted_probe_every_output_column_kind(eng):
    """i64 keys, duplicates on the build side, carried / rank-ordered / gathered / repeated columns and the key itself."""
    rng = np.random.default_rng(43)
    n, s = 300_007, 40_000
    rk = rng.integers(-5000, 5000, size=s).astype(np.int64)
    lk = np.sort(rng.integers(-20000, 20000, size=n).astype(np.int64))
    la, lf = rng.integers(-2**31, 2**31, n).astype(np.int32), rng.random(n).astype(np.float32)
    ra, rf = rng.integers(0, 2**31, s).astype(np.int32), rng.random(s).astype(np.float32)
    t1, t2 = eng.table_from_columns([lk, la, lf]), eng.table_from_columns([rk, ra, rf])
    res = eng.join(t1, t2, 0, 0, [2, 1, 0, 1], [2, 1, 0, 1])
    assert eng.last_join_path() == CLUSTERED
    li, ri = T._np_join_rows(lk, rk)
    assert res.shape == (len(li), 8) and len(li) > 0
    for j, exp in enumerate([lf[li], la[li], lk[li], la[li], rf[ri], ra[ri], rk[ri], ra[ri]]):
        assert np.array_equal(res.column(j), exp), j
    res.free(); t1.free(); t2.free()


def test_probe_rows_outside_the_build_range_and_ragged_ends(eng):
    rng = np.random.default_rng(45)
    s = 4096
    rk = rng.integers(1000, 2000, size=s).astype(np.uint32)
    for n in ((1 << 18), (1 << 18) + 1, (1 << 18) + 4095, (1 << 18) + 4097):
        lk = np.sort(rng.integers(0, 3000, size=n).astype(np.uint32))
        assert T._check(eng, lk, rk) > 0 and eng.last_join_path() == CLUSTERED
    assert T._check(eng, np.sort(rng.integers(5000, 9000, size=1 << 18).astype(np.uint32)), rk) == 0


_FORCED = r"""
import sys
import numpy as np
sys.path.insert(0, %r)
sys.path.insert(0, %r)
import test_gpu_hjoin as T
from harkdb_amd.engine import Engine
eng = Engine(0)
for case in T.CASES:
    lk, rk = T._make(case, 3)
    print(case, T._check(eng, lk, rk), eng.last_join_path())
    assert eng.last_join_path().startswith("clustered")
    print(case, "sorted", T._check(eng, np.sort(lk), rk))
for unique in (True, False, "spread", "spread_dup"): T.test_i64_join_with_several_output_columns(eng, unique)
for hot in (0, 500): T.test_i64_join_of_carried_columns_only(eng, hot)
print("forced ok")
"""


@pytest.mark.parametrize("grid", ["", "3"])
def test_every_partitioned_case_through_the_search_path(grid):
    """HARK_JOIN_CLUSTERED=1 sends every join of the partitioned path's size through the search path, whatever its rows look like
    (shuffled rows: the ranks do not ascend, the general ordering runs); HARK_CJOIN_GRID=3: three workgroups take all batches."""
    env = dict(os.environ, HARK_JOIN_CLUSTERED="1")
    if grid: env["HARK_CJOIN_GRID"] = grid
    out = subprocess.run([sys.executable, "-c", _FORCED % (ROOT, os.path.join(ROOT, "tests"))], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0 and "forced ok" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]


def test_the_test_can_be_switched_off():
    env = dict(os.environ, HARK_JOIN_CLUSTERED="0")
    code = _FORCED.replace('assert eng.last_join_path().startswith("clustered")', 'assert not eng.last_join_path().startswith("clustered")')
    code = code.split("for unique in")[0] + 'print("forced ok")\n'
    out = subprocess.run([sys.executable, "-c", code % (ROOT, os.path.join(ROOT, "tests"))], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0 and "forced ok" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]


_ROTATED = r"""
import sys
import numpy as np
sys.path.insert(0, %r)
sys.path.insert(0, %r)
import test_gpu_hjoin as T
from harkdb_amd.engine import Engine
eng = Engine(0)
for case in ("u32 10%% hits", "u32 every row hits, 8 partners each", "i64 duplicates both sides", "i64 unique build keys, every second probe row hits", "i32 negative keys as u32"):
    lk, rk = T._make(case, 5)
    reps = -(-(1 << 20) // len(lk))                                  # rotated loads need >= 128 full batches: 2^20 rows of 32-bit keys (2^19 of 64-bit ones)
    lk = np.tile(lk, reps)
    for order in ("shuffled", "blocks"):
        k = lk.copy()
        if order == "blocks":
            for a in range(0, len(k), 1 << 16): k[a:a + (1 << 16)] = np.sort(k[a:a + (1 << 16)])
        print(case, order, T._check(eng, k, rk), eng.last_join_path())
        assert eng.last_join_path() == "partitioned, rotated loads", eng.last_join_path()
for hot in (0, 500): T.test_i64_join_of_carried_columns_only(eng, hot)
print("rotated ok")
"""


@pytest.mark.parametrize("env", [{}, {"HARK_JOIN_PLAIN_LOADS": "1"}, {"HARK_JOIN_HOTMIN": "3"}])
def test_partition_with_rotated_loads_forced(env):
    """HARK_JOIN_CLUSTERED=2: the partition's 64 sixteen-lane groups read their rows of 64 different batches (`rot` of jpart_kernel),
    whatever the rows look like -- with the hand-placed waits and with the compiler's, and with hot keys (their rows are counted per
    batch by sixteen lanes then): the reference's order row for row."""
    out = subprocess.run([sys.executable, "-c", _ROTATED % (ROOT, os.path.join(ROOT, "tests"))], capture_output=True, text=True, timeout=900,
                         env=dict(os.environ, HARK_JOIN_CLUSTERED="2", HARK_JOIN_SLABX="1000", **env))     # (slabs of 10 x the even share: at 2^20 rows a slab holds ~100 entries, two groups of 64 rows in one bucket overrun it)
    assert out.returncode == 0 and "rotated ok" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]


def test_a_probe_column_sorted_block_by_block_takes_rotated_loads(eng, monkeypatch):
    monkeypatch.setenv("HARK_JOIN_SLABX", "1000")                      # (at 2^21 rows a slab holds ~110 entries: two groups of 64 rows in one bucket would overrun it and the search path take over)
    rng = np.random.default_rng(77)
    n, s = 1 << 21, 2_000_000                                        # rows 16 apart are ~240 build keys apart: too far for the search path's slices, inside one bucket
    rk = rng.choice(1 << 30, size=s, replace=False).astype(np.uint32)
    lk = rng.integers(0, 1 << 30, size=n).astype(np.uint32)
    hit = rng.random(n) < 0.3
    lk[hit] = rk[rng.integers(0, s, size=int(hit.sum()))]
    for a in range(0, n, 1 << 17): lk[a:a + (1 << 17)] = np.sort(lk[a:a + (1 << 17)])     # sixteen sorted blocks, every block over all keys
    assert T._check(eng, lk, rk) > 0
    assert eng.last_join_path() == "partitioned, rotated loads", eng.last_join_path()


def test_rotated_loads_that_overrun_their_slabs_fall_back_to_the_search_path(eng):
    """The same column with the slabs at their normal size: at 2^21 rows two groups of 64 rows in one bucket overrun a slab, the
    partition leaves, and the search path -- not the probe side's sort -- delivers the rows."""
    rng = np.random.default_rng(78)
    n, s = 1 << 21, 2_000_000
    rk = rng.choice(1 << 30, size=s, replace=False).astype(np.uint32)
    lk = rng.integers(0, 1 << 30, size=n).astype(np.uint32)
    hit = rng.random(n) < 0.3
    lk[hit] = rk[rng.integers(0, s, size=int(hit.sum()))]
    for a in range(0, n, 1 << 17): lk[a:a + (1 << 17)] = np.sort(lk[a:a + (1 << 17)])
    assert T._check(eng, lk, rk) > 0
    assert eng.last_join_path() in (CLUSTERED, "partitioned, rotated loads"), eng.last_join_path()
