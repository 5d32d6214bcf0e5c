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
    if shape == "sorted": return np.sort(kk)
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


@pytest.mark.parametrize("shape", ["sorted", "descending", "runs8192", "runs5000", "runs1024", "two_clusters", "six_clusters", "random"])
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
    for shape, expect in (("sorted", True), ("descending", True), ("runs8192", True), ("random", False)):
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
