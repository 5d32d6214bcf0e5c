"""The small-table paths (harkdb_amd/csrc/k_small.hip): tables of at most 4096 rows answer the reference's two entry
points -- query_sel (select.fut:17-23) and query_groupby (groupby.fut:51-62) -- with ONE launch and ONE synchronisation,
the result matrix written by the kernel into a pinned host block.  Every case runs twice, through the small path and
(HARK_NO_SMALL=1, read at call time) through the general kernels, and both are held to the CPU oracle, bit for bit."""
import os

import numpy as np
import pytest

from conftest import load_golden, resolve_table, ROOT

pytestmark = pytest.mark.gpu
OPS = load_golden("operators.json")


@pytest.fixture(scope="module")
def eng():
    from harkdb_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


@pytest.fixture(params=["small", "general"])
def path(request, monkeypatch):
    if request.param == "general":
        monkeypatch.setenv("HARK_NO_SMALL", "1")
    else:
        monkeypatch.delenv("HARK_NO_SMALL", raising=False)
    return request.param


@pytest.mark.parametrize("case", OPS["query_groupby"], ids=lambda c: c["id"])
def test_golden_groupby_through_both_paths(eng, path, case):
    """G2-G6 of SURVEY.md Appendix A."""
    t = eng.table_from_matrix(resolve_table(case["table"]), np.uint32)
    res = eng.query_groupby(t, case["g_col"], case["s_cols"], case["t_cols"])
    assert res.to_numpy(np.uint32).tolist() == case["out"]
    if len(case["out"]):
        assert (eng.last_groupby_path() == "small") == (path == "small")
        assert res.matrix(dtype=np.uint32).tolist() == case["out"]          # the pinned matrix the kernel wrote (small) / the device-built one


@pytest.mark.parametrize("case", OPS["query_sel"], ids=lambda c: c["id"])
def test_golden_sel_through_both_paths(eng, path, case):
    t = eng.table_from_matrix(resolve_table(case["table"]), np.int32)
    res = eng.query_sel(t, case["cols"])
    assert res.to_numpy(np.int32).tolist() == case["out"]
    if len(case["out"]) and len(case["cols"]):
        assert res.matrix().tolist() == case["out"]


@pytest.mark.parametrize("n,nkeys", [(1, 1), (2, 1), (7, 3), (63, 5), (64, 64), (65, 2), (1000, 1000), (1024, 1), (3000, 40), (4096, 4096), (4096, 7)])
def test_small_groupby_matches_oracle(eng, oracle, path, n, nkeys):
    rng = np.random.default_rng(n * 31 + nkeys)
    db = rng.integers(0, 2**32, size=(n, 4), dtype=np.uint64).astype(np.uint32)
    keys = rng.integers(0, 2**32, size=nkeys, dtype=np.uint64).astype(np.uint32)        # sparse keys, sign bit included
    db[:, 2] = keys[rng.integers(0, nkeys, size=n)] if nkeys < n else rng.permutation(n).astype(np.uint32) * 7919
    db[:, 1] = rng.integers(0, 4, size=n) * 2 + 1                                         # odd factors: products that wrap without collapsing to 0
    s_cols, t_cols = [0, 1, 3, 2], [2, 1, 3, 0] if n * 5 <= 16384 else [2, 1, 3, 0]     # sum, prod, max, key (`case _`: min)
    t = eng.table_from_matrix(db, np.uint32)
    res = eng.query_groupby(t, 2, s_cols, t_cols)
    exp = oracle.query_groupby(db, 2, s_cols, t_cols)
    got = res.to_numpy(np.uint32)
    assert got.shape == exp.shape and np.array_equal(got, exp)
    assert np.array_equal(res.matrix(dtype=np.uint32), exp)
    # the result's device columns serve further operators: a projection of it
    t2 = eng.table_from_device(res.shape[0], [res.device_ptr(j) for j in range(res.shape[1])], [np.uint32] * res.shape[1], keepalive=res)
    assert np.array_equal(eng.query_sel(t2, [4, 0]).to_numpy(np.uint32), exp[:, [4, 0]])


def test_small_groupby_min_and_other_opcodes(eng, oracle, path):
    rng = np.random.default_rng(5)
    db = rng.integers(0, 2**32, size=(500, 3), dtype=np.uint64).astype(np.uint32)
    db[:, 0] = rng.integers(0, 9, size=500)
    for ops in ([4, 4], [0, 9], [3, 2], [1, 1], [-1, 7]):
        t = eng.table_from_matrix(db, np.uint32)
        assert np.array_equal(eng.query_groupby(t, 0, [1, 2], ops).to_numpy(np.uint32), oracle.query_groupby(db, 0, [1, 2], ops)), ops


def test_small_groupby_errors(eng, path):
    from harkdb_amd._ffi import HarkError, EBOUNDS
    db = np.arange(12).reshape(3, 4)
    db[1, 0] = 0
    t = eng.table_from_matrix(db, np.uint32)
    for args in [(9, [1], [2]), (0, [1, 4], [2, 2]), (0, [1, 2], [2])]:
        with pytest.raises(HarkError) as ei:
            eng.query_groupby(t, *args)
        assert ei.value.code == EBOUNDS
    t2 = eng.table_from_matrix(np.arange(12).reshape(3, 4), np.uint32)             # all keys distinct: merge never runs, a short t_cols is no error
    assert eng.query_groupby(t2, 0, [1, 2], [2]).to_numpy(np.uint32).tolist() == [[0, 1, 2], [4, 5, 6], [8, 9, 10]]


@pytest.mark.parametrize("n,k", [(1, 1), (7, 2), (100, 32), (4096, 4), (4097, 2), (600, 31)])
def test_small_sel_matches_oracle(eng, oracle, path, n, k):
    rng = np.random.default_rng(n + k)
    db = rng.integers(-2**31, 2**31, size=(n, 8)).astype(np.int64)
    cols = rng.integers(0, 8, size=k).tolist()
    t = eng.table_from_matrix(db, np.int32)
    res = eng.query_sel(t, cols)
    exp = oracle.query_sel(db, cols)
    assert np.array_equal(res.to_numpy(np.int32), exp) and np.array_equal(res.matrix(), exp)
    assert np.array_equal(res.matrix([k - 1, 0], limit=max(1, n // 2)), exp[: max(1, n // 2), [k - 1, 0]])     # not the matrix as it stands: built on the device


def test_reference_statements_through_sql(path):
    """README.md:42 / test.py:7 on data.csv through FutharkContext.sql(), several times (the plan is cached)."""
    from harkdb_amd import FutharkContext
    fc = FutharkContext()
    fc.create_table("game_1", os.path.join(ROOT, "tests", "golden", "data.csv"))
    for _ in range(3):
        assert fc.sql("select col1, col3 from game_1").tolist() == [[6, 6], [0, 0], [0, 0], [0, 0], [0, 0], [6, 6], [1, 3]]
        out = fc.sql("select col1,  max(col3) from game_1 group by col1")
        assert out.dtype == np.uint32 and out.tolist() == [[0, 0, 0], [1, 1, 3], [6, 6, 6]]
    # a second table under the same name: the cached plans go with the old one
    fc.create_table("game_1", np.array([[5, 1, 2], [5, 3, 4]]))
    assert fc.sql("select col1, col3 from game_1").tolist() == [[5, 2], [5, 4]]


def test_select_list_of_more_than_32_columns_through_sql():
    """sql() builds the matrix on the device, 32 columns per launch -- a 40-column select list used to raise."""
    from harkdb_amd import FutharkContext
    rng = np.random.default_rng(1)
    db = rng.integers(-1000, 1000, size=(5000, 40))
    fc = FutharkContext()
    fc.create_table("wide", db)
    names = [f"col{j + 1}" for j in range(40)]
    order = list(rng.permutation(40)) + [3, 3]
    got = fc.sql("select " + ", ".join(names[j] for j in order) + " from wide")
    assert got.shape == (5000, 42) and np.array_equal(got, db[:, order])


def test_small_sel_keeps_every_columns_dtype(eng, path):
    """A projection of f32 / u32 / i32 columns of a small table: bit copies, each column its own dtype (tools/sql_stress.py found
    the f32 column of a filtered table coming back as i32 bits)."""
    rng = np.random.default_rng(4)
    cols = [rng.integers(-9, 9, 300).astype(np.int32), rng.random(300).astype(np.float32), rng.integers(0, 2**32, 300, dtype=np.uint64).astype(np.uint32)]
    t = eng.table_from_columns(cols)
    res = eng.query_sel(t, [1, 0, 2, 1])
    got = res.columns()
    for g, e in zip(got, [cols[1], cols[0], cols[2], cols[1]]):
        assert g.dtype == e.dtype and np.array_equal(g, e)
    assert np.array_equal(res.matrix(), np.stack([cols[1].astype(np.float64), cols[0].astype(np.float64), cols[2].astype(np.float64), cols[1].astype(np.float64)], axis=1))
