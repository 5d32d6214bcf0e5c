"""Property-based parity (hypothesis): random small tables, column lists and opcodes
through the C ABI against the oracle.  Bit-exact."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st, HealthCheck

pytestmark = pytest.mark.gpu

_eng = None


def engine():
    global _eng
    if _eng is None:
        from harkdb_amd.engine import Engine
        _eng = Engine(0)
    return _eng


COMMON = dict(deadline=None, max_examples=60, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])

tables = st.integers(0, 60).flatmap(lambda n: st.integers(1, 6).flatmap(lambda m: st.lists(
    st.lists(st.one_of(st.integers(0, 5), st.integers(0, 2**32 - 1), st.sampled_from([0, 1, 2**31, 2**32 - 1])), min_size=m, max_size=m),
    min_size=n, max_size=n).map(lambda rows: np.asarray(rows, dtype=np.uint64).reshape(n, m).astype(np.uint32))))


@settings(**COMMON)
@given(db=tables, data=st.data())
def test_query_sel_property(oracle, db, data):
    n, m = db.shape
    cols = data.draw(st.lists(st.integers(0, m - 1), min_size=0, max_size=5))
    eng = engine()
    t = eng.table_from_matrix(db.view(np.int32), np.int32)
    got = eng.query_sel(t, cols).to_numpy(np.int32) if cols else np.empty((n, 0), np.int32)
    assert np.array_equal(got, oracle.query_sel(db.view(np.int32), cols))


@settings(**COMMON)
@given(db=tables, data=st.data())
def test_query_groupby_property(oracle, db, data):
    n, m = db.shape
    g_col = data.draw(st.integers(0, m - 1))
    s_cols = data.draw(st.lists(st.integers(0, m - 1), min_size=0, max_size=4))
    t_cols = data.draw(st.lists(st.integers(0, 6), min_size=len(s_cols), max_size=len(s_cols) + 1))
    eng = engine()
    t = eng.table_from_matrix(db, np.uint32)
    got = eng.query_groupby(t, g_col, s_cols, t_cols).to_numpy(np.uint32)
    exp = oracle.query_groupby(db, g_col, s_cols, t_cols)
    assert got.shape == exp.shape and np.array_equal(got, exp)


@settings(**COMMON)
@given(a=tables, b=tables, data=st.data())
def test_join_property(oracle, a, b, data):
    c1 = data.draw(st.integers(0, a.shape[1] - 1)); c2 = data.draw(st.integers(0, b.shape[1] - 1))
    cols1 = data.draw(st.lists(st.integers(0, a.shape[1] - 1), min_size=0, max_size=3))
    cols2 = data.draw(st.lists(st.integers(0, b.shape[1] - 1), min_size=0, max_size=3))
    eng = engine()
    t1, t2 = eng.table_from_matrix(a, np.uint32), eng.table_from_matrix(b, np.uint32)
    res = eng.join(t1, t2, c1, c2, cols1, cols2)
    exp = oracle.join(a, b, c1, c2, cols1, cols2)
    assert res.shape[0] == exp.shape[0]
    if cols1 or cols2:
        assert np.array_equal(res.to_numpy(np.uint32), exp)


@settings(**COMMON)
@given(vals=st.lists(st.one_of(st.floats(-10, 10, width=32), st.sampled_from([0.0, -0.0, 0.5, float("inf"), float("-inf"), float("nan")])), min_size=0, max_size=300),
       thr=st.one_of(st.floats(-10, 10, width=32), st.just(0.5)), cmp=st.sampled_from([">", ">=", "<", "<=", "=", "!="]))
def test_filter_f32_property(oracle, vals, thr, cmp):
    col = np.asarray(vals, dtype=np.float32)
    eng = engine()
    t = eng.table_from_columns([col, np.arange(col.size, dtype=np.int32)])
    res = eng.filter_sel(t, 0, cmp, thr, [1])
    idx = oracle.filter_indices(col, cmp, np.float32(thr))
    assert np.array_equal(res.column(0), idx) and np.array_equal(res.column(1), idx.astype(np.int32))


@settings(**COMMON)
@given(keys=st.lists(st.integers(0, 40), min_size=0, max_size=400), data=st.data())
def test_fused_dense_groupby_property(oracle, keys, data):
    from harkdb_amd.engine import FgbPlan
    n = len(keys)
    G = data.draw(st.sampled_from([41, 64, 5000, 100_000]))
    algo = data.draw(st.sampled_from([0, 1, 2, 3])) if G * 12 <= 96 * 1024 else data.draw(st.sampled_from([0, 2, 3]))
    rng = np.random.default_rng(n + G)
    kk = np.asarray(keys, dtype=np.int32)
    pp = rng.random(n, dtype=np.float32)
    vv = rng.integers(0, 16, size=n).astype(np.float32)
    eng = engine()
    p, k, v = eng.alloc(max(n, 4) * 4), eng.alloc(max(n, 4) * 4), eng.alloc(max(n, 4) * 4)
    if n:
        eng.upload(p, pp); eng.upload(k, kk); eng.upload(v, vv)
    s, c = eng.alloc(G * 4), eng.alloc(G * 8)
    plan = FgbPlan(eng, max(n, 1), G, algo=algo)
    plan.run(p, ">", 0.5, k, v, n)
    plan.finish(s, c)
    s32, _, cnt = oracle.filter_groupby_dense_f32(pp, kk, vv, ">", 0.5, G)
    ok = np.array_equal(eng.download(c, G, np.int64), cnt) and np.array_equal(eng.download(s, G, np.float32), s32)
    plan.free()
    for ptr in (p, k, v, s, c):
        eng.free(ptr)
    assert ok
