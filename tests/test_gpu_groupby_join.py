"""GPU parity of the reference entry points query_groupby (groupby.fut) and
join (join.fut), plus SORT BY, against the CPU oracle: bit-exact."""
import numpy as np
import pytest

from conftest import load_golden, resolve_table, expand_runs

pytestmark = pytest.mark.gpu
OPS = load_golden("operators.json")


@pytest.fixture(scope="module")
def eng():
    from harkdb_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


@pytest.mark.parametrize("case", OPS["query_groupby"], ids=lambda c: c["id"])
def test_query_groupby_golden(eng, case):
    t = eng.table_from_matrix(resolve_table(case["table"]), np.uint32)
    out = eng.query_groupby(t, case["g_col"], case["s_cols"], case["t_cols"]).to_numpy(np.uint32)
    assert out.tolist() == case["out"]


@pytest.mark.parametrize("n,nkeys", [(1, 1), (63, 5), (64, 64), (4097, 3), (100_003, 1000), (300_000, 1), (262_144, 262_144)])
def test_query_groupby_matches_oracle(eng, oracle, n, nkeys):
    rng = np.random.default_rng(n * 31 + nkeys)
    db = rng.integers(0, 2**32, size=(n, 6), dtype=np.uint64).astype(np.uint32)
    keys = rng.integers(0, 2**32, size=nkeys, dtype=np.uint64).astype(np.uint32)   # sparse keys, sign bit included
    db[:, 2] = keys[rng.integers(0, nkeys, size=n)] if nkeys < n else rng.permutation(n).astype(np.uint32) * 7919
    s_cols, t_cols = [0, 1, 3, 4, 5, 2], [2, 1, 3, 4, 9, 0]      # sum, prod, max, min, `case _` (min), key
    t = eng.table_from_matrix(db, np.uint32)
    got = eng.query_groupby(t, 2, s_cols, t_cols).to_numpy(np.uint32)
    exp = oracle.query_groupby(db, 2, s_cols, t_cols)
    assert got.shape == exp.shape and np.array_equal(got, exp)


@pytest.mark.parametrize("n,nkeys,naggs", [(50_000, 40_000, 8), (50_000, 300, 9), (120_000, 1, 12)])
def test_query_groupby_sort_path_many_aggregates(eng, oracle, n, nkeys, naggs):
    """The sort path's two-pass tail reduces up to eight aggregates in one sweep; more take the separate reductions.  Mostly
    distinct keys, few keys (runs across waves and tiles: 32-bit atomics on the results) and one key."""
    rng = np.random.default_rng(n + nkeys + naggs)
    db = rng.integers(0, 2**32, size=(n, 4), dtype=np.uint64).astype(np.uint32)
    db[:, 0] = rng.integers(0, 2**32, size=nkeys, dtype=np.uint64).astype(np.uint32)[rng.integers(0, nkeys, size=n)]
    db[:, 3] = rng.integers(0, 4, size=n) * 2 + 1
    s_cols = [(1, 2, 3)[j % 3] for j in range(naggs)]
    t_cols = [(2, 3, 4, 1, 9)[j % 5] for j in range(naggs)]
    t = eng.table_from_matrix(db, np.uint32)
    got = eng.query_groupby(t, 0, s_cols, t_cols).to_numpy(np.uint32)
    exp = oracle.query_groupby(db, 0, s_cols, t_cols)
    assert got.shape == exp.shape and np.array_equal(got, exp)


@pytest.mark.parametrize("n,G", [(7, 7), (200_000, 5000), (300_000, 1 << 20), (1_000_000, 70_000)])
def test_query_groupby_dense_keys_fused_path(eng, oracle, n, G):
    """Keys below 2^21 take the fused dense kernels (LDS tables or partition + LDS),
    one pass per aggregate; result must still be the reference's, bit for bit."""
    rng = np.random.default_rng(n + G)
    db = rng.integers(0, 2**32, size=(n, 5), dtype=np.uint64).astype(np.uint32)
    db[:, 0] = rng.integers(0, G, size=n)
    db[:, 4] = rng.integers(0, 4, size=n) * 2 + 1                  # small odd factors: products stay interesting mod 2^32
    s_cols, t_cols = [1, 4, 2, 3, 0], [2, 1, 3, 4, 0]
    t = eng.table_from_matrix(db, np.uint32)
    got = eng.query_groupby(t, 0, s_cols, t_cols).to_numpy(np.uint32)
    exp = oracle.query_groupby(db, 0, s_cols, t_cols)
    assert got.shape == exp.shape and np.array_equal(got, exp)
    assert np.array_equal(eng.query_groupby(t, 0, [], []).to_numpy(np.uint32), exp[:, :1])   # key column only


@pytest.mark.parametrize("n,G", [(300_000, 1 << 17), (1_200_000, 1 << 20), (500_000, 20_000), (400_000, 3000)])
def test_query_groupby_several_aggregates_of_one_column(eng, oracle, n, G):
    """sum, max and min of ONE column come from one statistics pass that carries the column once (the 64-bit sum's low word
    is the reference's sum mod 2^32); a product and another column's aggregates keep their own passes.  Small G or few
    keys per bucket decline the pass -- the same table either way."""
    rng = np.random.default_rng(n % 977 + G)
    db = rng.integers(0, 2**32, size=(n, 4), dtype=np.uint64).astype(np.uint32)
    db[:, 0] = rng.integers(0, G, size=n)
    db[:, 3] = rng.integers(0, 4, size=n) * 2 + 1
    s_cols, t_cols = [1, 1, 1, 3, 2, 2, 1], [2, 3, 4, 1, 4, 3, 2]     # sum, max, min of col 1; prod of col 3; min, max of col 2; sum of col 1 again
    t = eng.table_from_matrix(db, np.uint32)
    got = eng.query_groupby(t, 0, s_cols, t_cols).to_numpy(np.uint32)
    exp = oracle.query_groupby(db, 0, s_cols, t_cols)
    assert got.shape == exp.shape and np.array_equal(got, exp)


@pytest.mark.parametrize("n,G,skew", [(1_300_000, 1 << 20, False), (700_001, 50_000, False), (900_000, 1 << 18, True), (300_000, 9000, False)])
def test_query_groupby_three_columns_in_one_pass(eng, oracle, n, G, skew):
    """sum / max / min of THREE different columns come from one triple pass (14-byte entries, every row survives), the next two
    from a pair pass; skewed keys (a third of the rows on one key) and a small G decline it -- the same table either way."""
    rng = np.random.default_rng(n % 991 + G)
    db = rng.integers(0, 2**32, size=(n, 6), dtype=np.uint64).astype(np.uint32)
    db[:, 0] = rng.integers(0, G, size=n)
    if skew:
        db[: n // 3, 0] = 4242
    t = eng.table_from_matrix(db, np.uint32)
    for s_cols, t_cols in (([1, 2, 3], [2, 3, 4]), ([5, 4, 3, 2, 1], [4, 3, 4, 3, 2]), ([1, 2, 3, 4, 5, 1], [3, 3, 4, 4, 1, 9])):
        got = eng.query_groupby(t, 0, s_cols, t_cols).to_numpy(np.uint32)
        exp = oracle.query_groupby(db, 0, s_cols, t_cols)
        assert got.shape == exp.shape and np.array_equal(got, exp), (s_cols, t_cols)


@pytest.mark.parametrize("n,ndistinct", [(400_000, 1000), (2_000_000, 500_000), (3_000_000, 3_000_000), (600_000, 3)])
def test_query_groupby_sparse_keys_hash_path(eng, oracle, n, ndistinct):
    """Sparse u32 keys (anywhere in [0, 2^32)) with >= 2^18 rows: hash partition + LDS hash
    tables (several rounds when a bucket holds more distinct keys than its table); heavy skew
    overflows the slabs and falls back to the sort-based path.  Same result either way."""
    rng = np.random.default_rng(n + ndistinct)
    pool = rng.integers(0, 2**32, size=ndistinct, dtype=np.uint64).astype(np.uint32)
    pool[:3] = [0, 0xFFFFFFFF, 0x80000000]                         # the extreme keys are ordinary keys
    db = rng.integers(0, 2**32, size=(n, 4), dtype=np.uint64).astype(np.uint32)
    db[:, 1] = pool[rng.integers(0, ndistinct, size=n)] if ndistinct < n else rng.permutation(pool)
    db[:, 3] = rng.integers(0, 3, size=n) * 2 + 1
    s_cols, t_cols = [0, 2, 3, 0, 0], [2, 3, 1, 4, 3]              # (the aggregates of column 0 share one hash partition)
    t = eng.table_from_matrix(db, np.uint32)
    got = eng.query_groupby(t, 1, s_cols, t_cols).to_numpy(np.uint32)
    exp = oracle.query_groupby(db, 1, s_cols, t_cols)
    assert got.shape == exp.shape and np.array_equal(got, exp)


@pytest.mark.parametrize("n,ndistinct", [(500_000, 70_000), (2_500_000, 2_200_000), (4_000_000, 4_000_000)])
def test_query_groupby_sparse_keys_several_operators_of_one_column(eng, oracle, n, ndistinct):
    """Two or three operators of ONE column over sparse keys come from one consumer pass (12- / 16-byte table entries, one
    sort of the result keys), a fourth takes the next pass; products wrap mod 2^32 through an LDS compare-and-swap; more
    distinct keys per bucket than a table holds take several rounds.  Bit for bit the oracle's table."""
    rng = np.random.default_rng(n % 983 + ndistinct)
    pool = rng.integers(0, 2**32, size=ndistinct, dtype=np.uint64).astype(np.uint32)
    db = rng.integers(0, 2**32, size=(n, 3), dtype=np.uint64).astype(np.uint32)
    db[:, 0] = pool[rng.integers(0, ndistinct, size=n)] if ndistinct < n else rng.permutation(pool)
    db[:, 2] = rng.integers(0, 3, size=n) * 2 + 1
    t = eng.table_from_matrix(db, np.uint32)
    for s_cols, t_cols in (([1, 1], [2, 3]), ([2, 1, 2, 1, 2, 2], [1, 4, 2, 2, 3, 4]), ([1, 1, 1, 1], [3, 4, 2, 3])):
        got = eng.query_groupby(t, 0, s_cols, t_cols).to_numpy(np.uint32)
        assert eng.last_groupby_path() == "hash"
        exp = oracle.query_groupby(db, 0, s_cols, t_cols)
        assert got.shape == exp.shape and np.array_equal(got, exp), (s_cols, t_cols)


def test_query_groupby_int32_table_is_viewed_as_u32(eng, oracle):
    """The Python layer uploads int32 columns; groupby.fut:51 reads them as u32."""
    rng = np.random.default_rng(3)
    db = rng.integers(-50, 50, size=(5000, 3)).astype(np.int64)
    t = eng.table_from_matrix(db, np.int32)
    got = eng.query_groupby(t, 0, [1, 2], [3, 2]).to_numpy(np.uint32)
    assert np.array_equal(got, oracle.query_groupby(db, 0, [1, 2], [3, 2]))
    assert got[-1, 0] == np.uint32(-1 & 0xFFFFFFFF)               # negative keys sort last


def test_query_groupby_errors(eng):
    from harkdb_amd._ffi import HarkError, EBOUNDS
    db = np.arange(12).reshape(3, 4)
    db[1, 0] = 0
    t = eng.table_from_matrix(db, np.uint32)
    for args in [(9, [1], [2]), (0, [1, 4], [2, 2]), (0, [1, 2], [2])]:
        with pytest.raises(HarkError) as ei:
            eng.query_groupby(t, *args)
        assert ei.value.code == EBOUNDS
    # all keys distinct: `merge` never runs, a short t_cols is not an error (as in the sequential fold)
    t2 = eng.table_from_matrix(np.arange(12).reshape(3, 4), np.uint32)
    assert eng.query_groupby(t2, 0, [1, 2], [2]).to_numpy(np.uint32).tolist() == [[0, 1, 2], [4, 5, 6], [8, 9, 10]]


@pytest.mark.parametrize("case", OPS["join"], ids=lambda c: c["id"])
def test_join_golden(eng, case):
    t1 = eng.table_from_matrix(resolve_table(case["db1"]), np.uint32)
    t2 = eng.table_from_matrix(resolve_table(case["db2"]), np.uint32)
    out = eng.join(t1, t2, case["col1"], case["col2"], case["cols1"], case["cols2"]).to_numpy(np.uint32)
    w = len(case["cols1"]) + len(case["cols2"])
    exp = expand_runs(case["out_runs"], w) if "out_runs" in case else np.asarray(case["out"], dtype=np.int64).reshape(-1, w)
    assert out.astype(np.int64).tolist() == exp.tolist()


@pytest.mark.parametrize("n,s,nkeys", [(1, 1, 1), (50, 70, 9), (5000, 300, 40), (3000, 3000, 3000), (20_000, 1000, 200_000)])
def test_join_matches_oracle(eng, oracle, n, s, nkeys):
    rng = np.random.default_rng(n + s)
    a = rng.integers(0, 2**32, size=(n, 3), dtype=np.uint64).astype(np.uint32)
    b = rng.integers(0, 2**32, size=(s, 4), dtype=np.uint64).astype(np.uint32)
    pool = rng.integers(0, 2**32, size=nkeys, dtype=np.uint64).astype(np.uint32)
    a[:, 1] = pool[rng.integers(0, nkeys, size=n)]
    b[:, 3] = pool[rng.integers(0, nkeys, size=s)]
    t1, t2 = eng.table_from_matrix(a, np.uint32), eng.table_from_matrix(b, np.uint32)
    got = eng.join(t1, t2, 1, 3, [0, 1, 2], [3, 0]).to_numpy(np.uint32)
    exp = oracle.join(a, b, 1, 3, [0, 1, 2], [3, 0])
    assert got.shape == exp.shape and np.array_equal(got, exp)


def test_join_i64_keys_extension(eng):
    """BASELINE configs[3] joins on an i64 key: same sort-merge, signed key order, any column width."""
    rng = np.random.default_rng(21)
    n, s = 20_000, 3_000
    pool = rng.integers(-2**62, 2**62, size=500)
    lk, rk = pool[rng.integers(0, 500, n)], pool[rng.integers(0, 500, s)]
    la, ra = rng.integers(0, 1000, n).astype(np.int32), rng.random(s).astype(np.float32)
    t1 = eng.table_from_columns([lk.astype(np.int64), la])
    t2 = eng.table_from_columns([ra, rk.astype(np.int64)])
    res = eng.join(t1, t2, 0, 1, [0, 1], [0, 1])
    got = [res.column(j) for j in range(4)]
    exp = []
    for key in np.unique(np.concatenate([lk, rk])):                # ascending signed keys, then left row, then right row
        li, ri = np.flatnonzero(lk == key), np.flatnonzero(rk == key)
        for i in li:
            for j in ri:
                exp.append((key, la[i], ra[j], rk[j]))
    assert len(exp) == res.shape[0]
    assert np.array_equal(got[0], np.asarray([e[0] for e in exp], dtype=np.int64)) and got[0].dtype == np.int64
    assert np.array_equal(got[1], np.asarray([e[1] for e in exp], dtype=np.int32))
    assert np.array_equal(got[2], np.asarray([e[2] for e in exp], dtype=np.float32))
    assert np.array_equal(got[3], got[0])


def test_join_empty_sides(eng):
    t1 = eng.table_from_matrix(np.zeros((0, 2)), np.uint32)
    t2 = eng.table_from_matrix(np.arange(6).reshape(3, 2), np.uint32)
    assert eng.join(t1, t2, 0, 0, [0], [1]).shape == (0, 2)
    assert eng.join(t2, t1, 0, 0, [0], [1]).shape == (0, 2)


@pytest.mark.parametrize("dtype", [np.uint32, np.int32, np.float32, np.int64])
@pytest.mark.parametrize("descending", [False, True])
@pytest.mark.parametrize("n", [0, 1, 4096, 150_001])
def test_sort_is_stable_and_ordered(eng, oracle, dtype, descending, n):
    rng = np.random.default_rng(n + 5)
    if dtype == np.float32:
        key = (rng.standard_normal(n) * 3).round(1).astype(np.float32)
    elif dtype == np.int64:
        key = rng.integers(-5, 5, size=n) * (2**33) + rng.integers(-2, 2, size=n)
    elif dtype == np.int32:
        key = rng.integers(-300, 300, size=n).astype(np.int32)
    else:
        key = rng.integers(0, 2**32, size=n, dtype=np.uint64).astype(np.uint32) & np.uint32(0xFF0000FF)
    rowid = np.arange(n, dtype=np.int32)
    t = eng.table_from_columns([np.ascontiguousarray(key.astype(dtype)), rowid])
    res = eng.sort(t, 0, [0, 1], descending=descending)
    k2, r2 = res.column(0), res.column(1)
    perm = np.argsort(-key.astype(np.float64) if descending and dtype != np.int64 else (key if not descending else -key), kind="stable")
    assert np.array_equal(r2, rowid[perm]) and np.array_equal(k2, key.astype(dtype)[perm])
    if dtype == np.uint32 and not descending:
        assert np.array_equal(r2, oracle.argsort_u32(key).astype(np.int32))    # the reference's 32-pass sort order


@pytest.mark.parametrize("n", [4096, 6144, 100_003, 1_300_001])
@pytest.mark.parametrize("kind", ["spread", "runs", "narrow_hi", "negative_cluster"])
@pytest.mark.parametrize("descending", [False, True])
def test_i64_sort_by_high_words(eng, n, kind, descending):
    """i64 keys whose high words differ travel through the passes as 16-byte tuples (sorted by the top bits of the high
    word, runs of equal prefixes fixed up in registers); a run longer than 16 falls back to the eight-pass sort.  Stable
    against numpy either way."""
    rng = np.random.default_rng(n % 1000 + len(kind))
    if kind == "spread":
        key = rng.integers(-2**63, 2**63 - 1, size=n)
    elif kind == "runs":                   # equal keys (stability) and keys that differ below the sorted prefix only
        pool = rng.integers(-2**63, 2**63 - 1, size=max(n // 5, 1))
        key = pool[rng.integers(0, len(pool), size=n)] + rng.integers(0, 4, size=n) * rng.integers(0, 2, size=n)
    elif kind == "narrow_hi":              # the high words differ in their low byte only: one pass, long runs -> the fallback
        key = rng.integers(0, 2**40, size=n)
    else:
        key = -rng.integers(0, 2**50, size=n) - (rng.integers(0, 2, size=n) << 62)
    key = key.astype(np.int64)
    rowid = np.arange(n, dtype=np.int32)
    t = eng.table_from_columns([key, rowid])
    res = eng.sort(t, 0, [0, 1], descending=descending)
    # descending and stable: ascending by the complemented key (equal keys keep their order)
    perm = np.argsort(~key if descending else key, kind="stable")
    assert np.array_equal(res.column(1), rowid[perm]) and np.array_equal(res.column(0), key[perm])
    res.free(); t.free()


@pytest.mark.parametrize("cols", [[0], [1], [1, 0, 1, 0], [2, 1], [0, 1, 2], [2, 0, 0], []])
@pytest.mark.parametrize("descending", [False, True])
def test_i64_sort_projections(eng, cols, descending):
    """ORDER BY an i64 key: the sorted keys and ONE carried 4-byte column are the sort's own outputs (first use takes the
    buffer, a repeat copies it), anything else -- an 8-byte column, a second column -- is gathered through the row ids."""
    rng = np.random.default_rng(len(cols) * 2 + int(descending))
    n = 20_011
    key = rng.integers(-2**62, 2**62, size=n).astype(np.int64)
    key[rng.integers(0, n, size=n // 10)] = key[3]
    a, b = rng.integers(-2**31, 2**31, size=n).astype(np.int32), rng.integers(-2**62, 2**62, size=n).astype(np.int64)
    src = [key, a, b]
    t = eng.table_from_columns(src)
    res = eng.sort(t, 0, cols, descending=descending)
    perm = np.argsort(~key if descending else key, kind="stable")
    assert res.shape == (n, len(cols))
    for j, c in enumerate(cols):
        assert np.array_equal(res.column(j), src[c][perm]), (j, c)
    res.free(); t.free()


@pytest.mark.parametrize("dtype", [np.uint32, np.int32, np.float32])
@pytest.mark.parametrize("cols", [[0], [1], [1, 0, 1, 0], [2, 1], [0, 1, 2], [2, 0, 0]])
@pytest.mark.parametrize("descending", [False, True])
def test_sort_projections(eng, dtype, cols, descending):
    """Every way ORDER BY produces its columns: the key from the sorted sort words, one carried
    payload column, gathers through the permutation, repeated columns."""
    rng = np.random.default_rng(17)
    n = 70_003
    if dtype == np.float32:
        key = rng.integers(-40, 40, size=n).astype(np.float32) / 4
        key[::97] = -0.0
        key[1::97] = 0.0
    else:
        key = rng.integers(0, 1 << 20, size=n).astype(dtype) - (dtype(1 << 19) if dtype == np.int32 else dtype(0))
    a = rng.integers(-1000, 1000, size=n).astype(np.int32)
    b = rng.random(n).astype(np.float32)
    t = eng.table_from_columns([key, a, b])
    res = eng.sort(t, 0, cols, descending=descending)
    order = key.astype(np.float64)
    perm = np.argsort(-order if descending else order, kind="stable")
    src = [key, a, b]
    for j, c in enumerate(cols):
        got, exp = res.column(j), src[c][perm]
        assert got.dtype == exp.dtype
        assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)), (cols, j)     # bit-exact, -0.0 stays -0.0


@pytest.mark.parametrize("dtype", [np.uint32, np.float32, np.int64])
def test_sort_constant_key_is_identity(eng, dtype):
    """All keys equal: no radix pass runs, rows keep table order."""
    n = 20_000
    key = np.full(n, 7, dtype=dtype)
    a = np.arange(n, dtype=np.int32)[::-1].copy()
    t = eng.table_from_columns([key, a])
    for cols in ([0, 1], [1], [1, 1, 0]):
        res = eng.sort(t, 0, cols, descending=True)
        for j, c in enumerate(cols):
            assert np.array_equal(res.column(j), [key, a][c])


def _np_join_rows(lk, rk):
    """(left row, right row) pairs of an inner join in the reference's order: ascending key, left row, right row."""
    ol, orr = np.argsort(lk, kind="stable"), np.argsort(rk, kind="stable")
    sl, sr = lk[ol], rk[orr]
    lb, ub = np.searchsorted(sr, sl, "left"), np.searchsorted(sr, sl, "right")
    cnt = ub - lb
    li = np.repeat(ol, cnt)
    within = np.arange(int(cnt.sum())) - np.repeat(np.cumsum(cnt) - cnt, cnt)
    ri = orr[np.repeat(lb, cnt) + within]
    return li, ri


def test_np_join_model_matches_oracle(oracle):
    rng = np.random.default_rng(2)
    a = rng.integers(0, 50, size=(400, 2)).astype(np.uint32)
    b = rng.integers(0, 50, size=(90, 2)).astype(np.uint32)
    li, ri = _np_join_rows(a[:, 0], b[:, 1])
    exp = oracle.join(a, b, 0, 1, [0, 1], [0, 1])
    assert np.array_equal(np.column_stack([a[li, 0], a[li, 1], b[ri, 0], b[ri, 1]]), exp)


@pytest.mark.parametrize("case", ["few hits", "no hits", "every probe row hits", "i64 few hits", "i64 no hits"])
def test_join_large_probe_side_prefilter(eng, case):
    """Probe side >= 2^20 rows and >= 4x the build side: the semi-join bitmap pre-filter runs (and is abandoned
    when more than half of the probe rows pass).  Results are identical row for row."""
    rng = np.random.default_rng(len(case))
    n, s = (1 << 20) + 12_345, 60_000
    wide = "i64" in case
    dt = np.int64 if wide else np.uint32
    hi = 2**62 if wide else 2**32
    rk = rng.integers(-hi if wide else 0, hi, size=s).astype(dt)
    if "every" in case:
        lk = rk[rng.integers(0, s, size=n)]
    elif "no hits" in case:
        lk = rng.integers(-hi if wide else 0, hi, size=n).astype(dt)
        lk = np.where(np.isin(lk, rk), lk + dt(1), lk)
        lk = lk[~np.isin(lk, rk)][:n]
        n = len(lk)
    else:
        lk = rng.integers(-hi if wide else 0, hi, size=n).astype(dt)
        hit = rng.random(n) < 0.07
        lk[hit] = rk[rng.integers(0, s, size=int(hit.sum()))]
        rk[:100] = rk[100:200]                                       # duplicate build keys: several partners per probe row
    la = np.arange(n, dtype=np.int32)
    rb = rng.random(s).astype(np.float32)
    t1, t2 = eng.table_from_columns([lk, la]), eng.table_from_columns([rb, rk])
    res = eng.join(t1, t2, 0, 1, [1, 0], [0, 1])
    li, ri = _np_join_rows(lk, rk)
    assert res.shape[0] == len(li)
    if len(li):
        assert np.array_equal(res.column(0), la[li]) and np.array_equal(res.column(1), lk[li])
        assert np.array_equal(res.column(2), rb[ri]) and np.array_equal(res.column(3), rk[ri])


@pytest.mark.parametrize("descending", [False, True])
def test_sort_f32_nan_and_infinities(eng, descending):
    """f32 keys: -inf < ... < -0.0 == +0.0 < ... < +inf < every NaN (numpy's order), stable; reversed for DESC."""
    rng = np.random.default_rng(9)
    n = 50_000
    key = rng.standard_normal(n).astype(np.float32)
    key[::41] = np.nan
    key[1::41] = -np.nan
    key[2::41] = np.inf
    key[3::41] = -np.inf
    key[4::41] = -0.0
    rowid = np.arange(n, dtype=np.int32)
    t = eng.table_from_columns([key, rowid])
    res = eng.sort(t, 0, [1, 0], descending=descending)
    rank = np.clip(key.astype(np.float64), -1e300, 1e300)
    rank[np.isnan(key)] = 2e300                                               # NaNs tie at the far end
    perm = np.argsort(-rank if descending else rank, kind="stable")
    assert np.array_equal(res.column(0), rowid[perm])
    assert np.array_equal(res.column(1).view(np.uint32), key[perm].view(np.uint32))


def test_a_rounds_hint_from_larger_tables_does_not_condemn_the_key_column(eng, oracle):
    """the hash consumers' tables differ in size (6144 keys per bucket and round with one or two operators, 4608
    with three), and the rounds a key column needed are remembered with the column.  ~2.7e6 distinct keys fit one round of the
    one-operator tables (5270 per bucket) but not of the three-operator ones: the remembered R = 1 must be doubled for the
    second statement, not answered with "unfit" -- which sent that and every later statement over the column to the sort path."""
    rng = np.random.default_rng(5)
    n, nd = 6_000_000, 2_700_000
    pool = rng.choice(1 << 32, nd, replace=False).astype(np.uint32)
    keys = pool[rng.integers(0, nd, n)]
    vals = rng.integers(0, 1 << 20, n).astype(np.uint32)
    db = np.stack([keys, vals], axis=1)
    t = eng.table_from_matrix(db, np.uint32)
    first = eng.query_groupby(t, 0, [1], [3])                              # max
    assert eng.last_groupby_path() == "hash"
    exp1 = oracle.query_groupby(db.astype(np.int64), 0, [1], [3])
    assert np.array_equal(first.to_numpy(np.uint32), exp1)
    second = eng.query_groupby(t, 0, [1, 1, 1], [2, 3, 4])                 # sum, max, min of one column: the three-operator tables
    assert eng.last_groupby_path() == "hash", "the remembered rounds must adapt, not flip the column to the sort path"
    exp2 = oracle.query_groupby(db.astype(np.int64), 0, [1, 1, 1], [2, 3, 4])
    assert np.array_equal(second.to_numpy(np.uint32), exp2)
    third = eng.query_groupby(t, 0, [1], [2])
    assert eng.last_groupby_path() == "hash" and np.array_equal(third.to_numpy(np.uint32), oracle.query_groupby(db.astype(np.int64), 0, [1], [2]))
    for r in (first, second, third):
        r.free()
    t.free()
