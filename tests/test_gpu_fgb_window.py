"""The fused WHERE -> GROUP BY on key columns that are SORTED or CLUSTERED by the key (k_fgb.hip, fgb_window_kernel: every
workgroup aggregates a contiguous stretch of the table in a window of consecutive keys in LDS that follows the keys) against
the CPU oracle (groupby.fut:8-58 restated) on the same inputs, through the C ABI."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from harkdb_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def _keys(shape, n, G, rng):
    kk = rng.integers(0, G, size=n).astype(np.int32)
    if shape == "random": return kk
    if shape == "blocks":                                              # sorted inside blocks of 2^16 rows, every block spans all keys (sorted files one behind the other)
        for a in range(0, n, 1 << 16): kk[a:a + (1 << 16)] = np.sort(kk[a:a + (1 << 16)])
        return kk
    if shape == "sorted": return np.sort(kk)
    if shape.startswith("noisy"):                                     # sorted, then that many per cent of the rows overwritten with keys from anywhere (late rows in a table kept in key order)
        s = np.sort(kk)
        idx = rng.integers(0, n, size=int(n * float(shape[5:]) / 100))
        s[idx] = kk[idx]
        return s
    if shape == "descending": return np.sort(kk)[::-1].copy()
    if shape.startswith("runs"):                                       # sorted, then runs of w rows shuffled as wholes (w need not divide a batch)
        w = int(shape[4:])
        s = np.sort(kk)
        m = n // w
        return np.concatenate([s[:m * w].reshape(m, w)[rng.permutation(m)].reshape(-1), s[m * w:]])
    if shape == "two_clusters":                                        # every batch of 4096 rows alternates between two far-apart stretches of keys
        s = np.sort(kk)
        half = n // 2
        out = np.empty(n, dtype=np.int32)
        out[0:2 * half:2] = s[:half]; out[1:2 * half:2] = s[half:2 * half]; out[2 * half:] = s[2 * half:]
        return out
    if shape == "six_clusters":                                        # ... six of them: more than the kernel takes turns for, the rest goes to global atomics
        s = np.sort(kk)
        m = n // 6
        out = np.empty(n, dtype=np.int32)
        for i in range(6): out[i:6 * m:6] = s[i * m:(i + 1) * m]
        out[6 * m:] = s[6 * m:]
        return out
    raise AssertionError(shape)


def _run(eng, oracle, kk, G, use_pred=True, count_only=False, cmp=">", thr=0.5, vop=None, seed=5, calls=1, **knobs):
    from harkdb_amd.engine import FgbPlan
    n = len(kk)
    rng = np.random.default_rng(seed)
    pp = rng.random(n, dtype=np.float32)
    vv = rng.integers(0, 16, size=n).astype(np.float32)
    p, k, v = eng.alloc(n * 4), eng.alloc(n * 4), eng.alloc(n * 4)
    eng.upload(p, pp); eng.upload(k, kk); eng.upload(v, vv)
    s, c = eng.alloc(G * 4), eng.alloc(G * 8)
    plan = FgbPlan(eng, n, G, algo=3, **knobs)
    for _ in range(calls):
        plan.run(p if use_pred else None, cmp, thr, k, None if count_only else v, n)
    plan.finish(s, c)
    got_s, got_c = eng.download(s, G, np.float32), eng.download(c, G, np.int64)
    s32, _, cnt = oracle.filter_groupby_dense_f32(pp if use_pred else None, kk, vv, cmp, thr, G)
    assert np.array_equal(got_c, cnt * calls)
    if not count_only: assert np.array_equal(got_s, s32 * np.float32(calls))          # integer-valued f32: exact in any order
    moves = plan.window_moves() if hasattr(plan, "window_moves") else None
    plan.free()
    for ptr in (p, k, v, s, c):
        eng.free(ptr)
    return moves


@pytest.mark.parametrize("shape", ["sorted", "descending", "noisy0.2", "noisy3", "noisy25", "runs8192", "runs5000", "runs1024", "two_clusters", "six_clusters", "random"])
@pytest.mark.parametrize("n,G", [(3_000_017, 1 << 20), (1_100_003, 300_000)])
def test_window_path_forced_on_every_shape(eng, oracle, shape, n, G):
    """window=1 sends any column through the window kernel: rows inside the window are added in LDS, a batch that straddles
    clusters takes several turns, what is left after the fourth goes to global atomics -- same bits as the oracle whatever the
    rows look like (shuffled keys: nearly every row goes the slow way)."""
    if shape == "random": n = 300_007
    _run(eng, oracle, _keys(shape, n, G, np.random.default_rng(3)), G, window=1)


@pytest.mark.parametrize("use_pred,count_only,cmp,thr", [(False, False, ">", 0.5), (True, True, ">", 0.5), (False, True, ">", 0.5), (True, False, "<=", 0.05), (True, False, "!=", 2.0)])
def test_window_path_predicates_and_count_only(eng, oracle, use_pred, count_only, cmp, thr):
    kk = _keys("sorted", 2_500_003, 1 << 20, np.random.default_rng(4))
    _run(eng, oracle, kk, 1 << 20, use_pred=use_pred, count_only=count_only, cmp=cmp, thr=thr, window=1)


def test_the_test_picks_the_window_for_sorted_keys_and_the_partition_for_shuffled_ones(eng, oracle):
    """window=0 (the default): fgb_cluster_test_kernel decides once per plan and column; the plan reports how often its
    windows moved (0: the partition path ran)."""
    from harkdb_amd.engine import FgbPlan
    rng = np.random.default_rng(6)
    n, G = 2_000_003, 1 << 17                                          # ~15 rows per key: rows 512 apart are ~34 keys apart when the column is sorted
    for shape, expect in (("sorted", True), ("descending", True), ("noisy2", True), ("runs8192", True), ("random", False)):
        kk = _keys(shape, n, G, rng)
        pp = rng.random(n, dtype=np.float32); vv = rng.integers(0, 16, size=n).astype(np.float32)
        p, k, v = eng.alloc(n * 4), eng.alloc(n * 4), eng.alloc(n * 4)
        eng.upload(p, pp); eng.upload(k, kk); eng.upload(v, vv)
        s, c = eng.alloc(G * 4), eng.alloc(G * 8)
        plan = FgbPlan(eng, n, G, timing=1)
        plan.run(p, ">", 0.5, k, v, n)
        plan.finish(s, c)
        s32, _, cnt = oracle.filter_groupby_dense_f32(pp, kk, vv, ">", 0.5, G)
        assert np.array_equal(eng.download(c, G, np.int64), cnt) and np.array_equal(eng.download(s, G, np.float32), s32)
        _, launches = plan.timing()
        assert (launches["consumer"] == 0) == expect, (shape, launches)      # the window path has no consumer pass
        plan.free()
        for ptr in (p, k, v, s, c):
            eng.free(ptr)


def test_accumulates_across_calls_and_plans_reset(eng, oracle):
    kk = _keys("sorted", 1_500_001, 1 << 20, np.random.default_rng(8))
    _run(eng, oracle, kk, 1 << 20, calls=3, window=1)


def test_key_out_of_range_is_bounds_error(eng):
    from harkdb_amd.engine import FgbPlan
    from harkdb_amd import _ffi
    n, G = 1_200_000, 1 << 20
    kk = np.sort(np.random.default_rng(9).integers(0, G, size=n)).astype(np.int32)
    kk[-5] = G + 7
    pp = np.ones(n, dtype=np.float32); vv = np.ones(n, dtype=np.float32)
    p, k, v = eng.alloc(n * 4), eng.alloc(n * 4), eng.alloc(n * 4)
    eng.upload(p, pp); eng.upload(k, kk); eng.upload(v, vv)
    plan = FgbPlan(eng, n, G, algo=3, window=1)
    plan.run(p, ">", 0.5, k, v, n)
    with pytest.raises(_ffi.HarkError):
        plan.check()
    plan.free()
    for ptr in (p, k, v):
        eng.free(ptr)


@pytest.mark.parametrize("n,G,order", [(1_300_000, 1 << 17, "sorted"), (1_300_000, 1 << 17, "descending"), (2_100_000, 1 << 18, "runs8192"), (1_200_000, 1 << 17, "random"), (1_400_000, 1 << 17, "blocks"), (1_300_000, 1 << 17, "noisy1")])
def test_reference_query_groupby_on_a_table_kept_in_key_order(eng, oracle, n, G, order):
    """query_groupby (main.fut:9) with several aggregates of one and of three columns on a table SORTED by its key column: the
    statistics / pair / triple passes run through the window kernel (fgb_windowx_kernel) -- same table as the oracle's."""
    rng = np.random.default_rng(n % 983 + G)
    db = rng.integers(0, 2**32, size=(n, 6), dtype=np.uint64).astype(np.uint32)
    db[:, 0] = _keys(order, n, G, rng).astype(np.uint32)
    db[:, 5] = rng.integers(0, 4, size=n) * 2 + 1
    t = eng.table_from_matrix(db, np.uint32)
    for s_cols, t_cols in (([1, 1, 1], [2, 3, 4]), ([1, 1, 1, 5, 2, 2, 1], [2, 3, 4, 1, 4, 3, 2]), ([1, 2, 3], [2, 3, 4]), ([4, 3, 2, 1], [4, 3, 4, 2]), ([1], [2]), ([], [])):
        got = eng.query_groupby(t, 0, s_cols, t_cols).to_numpy(np.uint32)
        exp = oracle.query_groupby(db, 0, s_cols, t_cols)
        assert got.shape == exp.shape and np.array_equal(got, exp), (s_cols, t_cols)
        assert eng.last_groupby_path() == "dense"
        assert eng.last_groupby_window() == (order not in ("random", "blocks")), (order, s_cols)
        assert eng.last_groupby_rotated() == (order == "blocks"), (order, s_cols)              # (blocks: the partition with rotated loads)
    t.free()


@pytest.mark.parametrize("order", ["sorted", "descending", "noisy1", "random", "blocks"])
def test_filter_groupby_entry_typed_aggregates_on_a_table_kept_in_key_order(eng, order):
    """hark_entry_filter_groupby: SUM / MAX / MIN / AVG / COUNT of f32, i32 and u32 columns under a predicate, keys sorted: the
    one-column statistics pass, the pair / triple passes and the single passes all take the window kernels; against pandas."""
    import pandas as pd
    rng = np.random.default_rng(17)
    n, G = 1_500_003, 1 << 17
    k = _keys(order, n, G, rng)
    df = pd.DataFrame({"p": rng.random(n).astype(np.float32), "k": k, "a": (rng.integers(-500, 500, n) / 4).astype(np.float32),
                       "i": rng.integers(-10**6, 10**6, n).astype(np.int32), "u": rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32),
                       "b": rng.normal(size=n).astype(np.float32)})
    t = eng.table_from_columns([df[c].to_numpy() for c in df.columns])
    aggs = [("sum", 2), ("max", 2), ("min", 2), ("avg", 2), ("count", 0), ("max", 3), ("min", 4), ("sum", 3), ("max", 5), ("avg", 3), ("sum", 4)]
    res = eng.filter_groupby(t, [(0, ">", 0.4)], 1, aggs)
    assert eng.last_groupby_path() == "dense" and eng.last_groupby_window() == (order not in ("random", "blocks")) and eng.last_groupby_rotated() == (order == "blocks")
    cols = res.columns()
    res.free()
    g = df[df.p > 0.4].groupby("k").agg(sa=("a", "sum"), mxa=("a", "max"), mna=("a", "min"), ava=("a", "mean"), n=("a", "count"), mxi=("i", "max"), mnu=("u", "min"),
                                        si=("i", "sum"), mxb=("b", "max"), avi=("i", "mean"), su=("u", "sum")).reset_index()
    exp = [g.k, g.sa, g.mxa, g.mna, g.ava, g.n, g.mxi, g.mnu, g.si, g.mxb, g.avi, g.su]
    assert len(cols) == len(exp)
    for j, (got, e) in enumerate(zip(cols, exp)):
        e = e.to_numpy()
        if got.dtype.kind == "f": assert np.allclose(got.astype(np.float64), e.astype(np.float64), rtol=2e-6, atol=1e-6), j
        else: assert np.array_equal(got.astype(np.int64), e.astype(np.int64)), j
    # ... and without a predicate, COUNT only
    res = eng.filter_groupby(t, [], 1, [("count", 0)])
    c2 = res.columns(); res.free()
    g2 = df.groupby("k").size().reset_index()
    assert np.array_equal(c2[0].astype(np.int64), g2.k.to_numpy()) and np.array_equal(c2[1].astype(np.int64), g2[0].to_numpy())
    t.free()


@pytest.mark.parametrize("shape", ["blocks", "sorted", "runs256", "random"])
@pytest.mark.parametrize("n,G,knobs", [(3_000_017, 1 << 20, {}), (2_200_003, 300_000, {"pairfmt": 2}), (2_500_001, 1 << 20, {"chunk_rows": 1 << 20})])
def test_partition_with_rotated_loads_forced_on_every_shape(eng, oracle, shape, n, G, knobs):
    """window=3: the producer's 64 sixteen-lane groups read their rows of 64 different batches (`rot` of fgb_part_kernel: for key
    columns whose neighbouring rows share a bucket -- sorted files one behind the other).  Every row is still read exactly once:
    same bits as the oracle, with and without a predicate, COUNT only, in chunks."""
    rng = np.random.default_rng(13)
    if shape == "blocks":                                              # sorted inside blocks of 2^18 rows, every block spans all keys
        kk = rng.integers(0, G, size=n).astype(np.int32)
        for a in range(0, n, 1 << 18): kk[a:a + (1 << 18)] = np.sort(kk[a:a + (1 << 18)])
    else:
        kk = _keys(shape, n, G, rng)
    _run(eng, oracle, kk, G, window=3, **knobs)
    _run(eng, oracle, kk, G, use_pred=False, window=3, **knobs)
    _run(eng, oracle, kk, G, count_only=True, window=3, **knobs)
