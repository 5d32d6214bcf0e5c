"""Column statistics cached with a table vs. columns the caller lent and then rewrote.

Tables are immutable (include/hark.h at hark_table_from_device; the reference re-passes the table per query,
FutharkContext.py:65,70, and so can never be stale).  hark_table_invalidate_stats is the caller's way to say "I rewrote a
borrowed column": afterwards every path sees the new contents -- the oracle's rows both times, and the LDS hash path the
second time although the first contents had sent the column to the sort path for good."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from harkdb_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def _borrowed(eng, cols):
    ptrs = [eng.alloc(c.nbytes) for c in cols]
    for p, c in zip(ptrs, cols):
        eng.upload(p, c)
    return eng.table_from_device(len(cols[0]), ptrs, [c.dtype for c in cols]), ptrs


def test_rewritten_borrowed_key_column_reference_groupby(eng, oracle):
    n = 400_000
    rng = np.random.default_rng(7)
    val = rng.integers(0, 2**32, size=n, dtype=np.uint64).astype(np.uint32)
    hot = np.full(n, 0xDEADBEEF, dtype=np.uint32)               # one sparse key: the hash partition's slab overflows -> sort path, sticky
    t, (pk, pv) = _borrowed(eng, [hot, val])
    got = eng.query_groupby(t, 0, [1, 1], [2, 3]).to_numpy(np.uint32)
    assert np.array_equal(got, oracle.query_groupby(np.stack([hot, val], 1), 0, [1, 1], [2, 3]))
    assert eng.last_groupby_path() == "sort"
    # the caller rewrites the column it lent: 5000 distinct keys spread over [0, 2^32)
    keys = (rng.integers(0, 5000, size=n).astype(np.uint64) * 2654435761 % (1 << 32)).astype(np.uint32)
    eng.upload(pk, keys)
    exp = oracle.query_groupby(np.stack([keys, val], 1), 0, [1, 1], [2, 3])
    # without the call the verdict is stale but only the PATH is: the rows are the new contents' rows
    assert np.array_equal(eng.query_groupby(t, 0, [1, 1], [2, 3]).to_numpy(np.uint32), exp)
    assert eng.last_groupby_path() == "sort"
    t.invalidate_stats()
    assert np.array_equal(eng.query_groupby(t, 0, [1, 1], [2, 3]).to_numpy(np.uint32), exp)
    assert eng.last_groupby_path() == "hash"
    t.free()
    eng.free(pk); eng.free(pv)


def test_rewritten_borrowed_key_column_dense_range(eng, oracle):
    from harkdb_amd._ffi import HarkError, EBOUNDS
    n = 300_000
    rng = np.random.default_rng(8)
    val = rng.integers(0, 1000, size=n).astype(np.uint32)
    k1 = rng.integers(0, 1000, size=n).astype(np.uint32)
    t, (pk, pv) = _borrowed(eng, [k1, val])
    assert np.array_equal(eng.query_groupby(t, 0, [1], [2]).to_numpy(np.uint32), oracle.query_groupby(np.stack([k1, val], 1), 0, [1], [2]))
    assert eng.last_groupby_path() == "dense" and eng.column_range(t, 0) == (int(k1.min()), int(k1.max()))
    k2 = rng.integers(0, 100_000, size=n).astype(np.uint32)
    eng.upload(pk, k2)
    with pytest.raises(HarkError) as ei:                        # the documented symptom of a stale range: loud, never silent
        eng.query_groupby(t, 0, [1], [2])
    assert ei.value.code == EBOUNDS
    t.invalidate_stats(0)
    assert eng.column_range(t, 0) == (int(k2.min()), int(k2.max()))
    assert np.array_equal(eng.query_groupby(t, 0, [1], [2]).to_numpy(np.uint32), oracle.query_groupby(np.stack([k2, val], 1), 0, [1], [2]))
    assert eng.last_groupby_path() == "dense"
    t.free()
    eng.free(pk); eng.free(pv)


def test_rewritten_borrowed_columns_multi_key_groupby():
    """GROUP BY on two keys folds them into a composite key with the cached ranges (hark_table_composite_key): after a
    rewrite + invalidate_table_stats the groups are the new contents' groups (pandas as the model)."""
    import pandas as pd
    from harkdb_amd import FutharkContext
    fc = FutharkContext(device=0, sql_mode=True)
    eng = fc.FutEnv
    n = 200_000
    rng = np.random.default_rng(9)

    def frame(hi_a, hi_b):
        return pd.DataFrame({"a": rng.integers(-5, hi_a, size=n).astype(np.int32), "b": rng.integers(0, hi_b, size=n).astype(np.int32),
                             "x": rng.integers(0, 100, size=n).astype(np.int32)})

    df = frame(5, 7)
    ptrs = [eng.alloc(n * 4) for _ in range(3)]
    for p, c in zip(ptrs, "abx"):
        eng.upload(p, df[c].to_numpy())
    fc.create_table_from_device("t", ["a", "b", "x"], ptrs, [np.int32] * 3, n)
    q = "select a, b, sum(x), count(*) from t group by a, b"

    def check(d):
        names, cols = fc.sql_columns(q)
        exp = d.groupby(["a", "b"], as_index=False).agg(s=("x", "sum"), c=("x", "size")).sort_values(["a", "b"])
        for got, want in zip(cols, (exp.a, exp.b, exp.s, exp.c)):
            assert np.array_equal(np.asarray(got, dtype=np.int64), want.to_numpy().astype(np.int64))

    check(df)
    df2 = frame(40, 300)                                          # wider key ranges than the cached ones
    for p, c in zip(ptrs, "abx"):
        eng.upload(p, df2[c].to_numpy())
    fc.invalidate_table_stats("t")
    check(df2)
    fc.drop_table("t")
    for p in ptrs:
        eng.free(p)


def test_sql_groupby_remembers_a_key_column_the_hash_path_does_not_fit(eng):
    """an unfiltered SQL GROUP BY over one hot sparse key overflows the hash partition (skew) and falls to the sort
    path; the verdict now stays with the key column (the reference entry did that already), so the second statement goes to
    the sort path directly -- same rows -- and hark_table_invalidate_stats brings the hash path back for rewritten keys.  A
    statement WITH a WHERE neither uses nor leaves a verdict."""
    n = 400_000
    rng = np.random.default_rng(11)
    hot = np.full(n, -559038737, dtype=np.int32)
    val = rng.integers(-1000, 1000, size=n).astype(np.int32)
    t, (pk, pv) = _borrowed(eng, [hot, val])
    for _ in range(2):
        r = eng.filter_groupby(t, [], 0, [("sum", 1), ("count", 0)])
        assert eng.last_groupby_path() == "sort"
        assert r.column(0).tolist() == [-559038737] and r.column(1).tolist() == [int(val.sum())] and r.column(2).tolist() == [n]
        r.free()
    keys = (rng.integers(0, 5000, size=n).astype(np.int64) * 2654435761 % (1 << 32)).astype(np.uint32).view(np.int32)
    eng.upload(pk, keys)
    t.invalidate_stats()
    r = eng.filter_groupby(t, [(1, ">=", 0)], 0, [("sum", 1), ("count", 0)])          # filtered: takes the hash path, leaves no verdict
    assert eng.last_groupby_path() == "hash"
    keep = val >= 0
    uk, inv = np.unique(keys[keep], return_inverse=True)
    assert np.array_equal(r.column(0), uk) and np.array_equal(r.column(1), np.bincount(inv, weights=val[keep]).astype(np.int64))
    r.free()
    r = eng.filter_groupby(t, [], 0, [("sum", 1), ("count", 0)])
    assert eng.last_groupby_path() == "hash"
    uk, inv = np.unique(keys, return_inverse=True)
    assert np.array_equal(r.column(0), uk) and np.array_equal(r.column(2), np.bincount(inv))
    r.free()
    t.free()
    eng.free(pk); eng.free(pv)
