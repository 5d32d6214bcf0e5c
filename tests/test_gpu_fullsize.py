"""BASELINE.json's full sizes, checked through size-independent properties
(the oracle cannot finish these sizes in seconds): linearity of the aggregate
in the predicate, checksums of checksums, complement counts, sortedness."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SEED = 0x4861726B4442


@pytest.fixture(scope="module")
def eng():
    from harkdb_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def test_c3_one_billion_rows_linearity_and_checksums(eng):
    """configs[2] + filter: 1e9 rows, 2^20 groups, integer-valued f32 values
    (sums exact in any order).  A(p > t) + A(p <= t) == A(no filter), per group,
    bit for bit; the count checksum equals an independent compaction count."""
    from harkdb_amd.engine import FgbPlan
    n, G = 1_000_000_000, 1 << 20
    p, k, v = eng.alloc(n * 4), eng.alloc(n * 4), eng.alloc(n * 4)
    eng.gen_columns(SEED, 0, n, G, True, p, k, v)
    s, c = eng.alloc(G * 4), eng.alloc(G * 8)
    plan = FgbPlan(eng, n, G)
    out = {}
    for name, pred, cmp in (("gt", p, ">"), ("le", p, "<="), ("all", None, ">")):
        plan.reset()
        plan.run(pred, cmp, 0.5, k, v, n)
        plan.finish(s, c)
        out[name] = (eng.download(s, G, np.float32).astype(np.float64), eng.download(c, G, np.int64))
    assert np.array_equal(out["gt"][1] + out["le"][1], out["all"][1])
    assert np.array_equal(out["gt"][0] + out["le"][0], out["all"][0])
    assert out["all"][1].sum() == n and out["all"][1].min() > 0
    # independent path for the survivor count: the WHERE compaction kernels
    t = eng.table_from_device(n, [p], [np.float32])
    assert eng.filter_sel(t, 0, ">", 0.5, [], want_row_index=False).shape[0] == out["gt"][1].sum()
    # generator property: v in {0..15} uniformly -> mean 7.5 within sampling error
    assert abs(out["all"][0].sum() / n - 7.5) < 1e-3
    plan.free()
    for ptr in (p, k, v, s, c):
        eng.free(ptr)


def test_c2_hundred_million_rows_filter_projection(eng):
    """configs[1]: WHERE + projection on a 1e8 x 8 f32 table."""
    n = 100_000_000
    cols = [eng.alloc(n * 4) for _ in range(8)]
    for j in range(0, 8, 2):
        eng.gen_columns(SEED + j, 0, n, 1 << 20, False, cols[j], None, cols[j + 1])
    t = eng.table_from_device(n, cols, [np.float32] * 8)
    res = eng.filter_sel(t, 1, ">", 0.5, [0, 2, 1], want_row_index=True)
    idx, c1 = res.column(0), res.column(3)
    assert np.all(np.diff(idx) > 0) and idx[0] >= 0 and idx[-1] < n      # ascending = order preserving
    assert c1.min() > 0.5                                                  # every survivor satisfies the predicate
    comp = eng.filter_sel(t, 1, "<=", 0.5, [], want_row_index=False)
    assert res.shape[0] + comp.shape[0] == n                               # complement
    # projected cells are the table's cells at the surviving rows (spot check by regenerating on the host)
    from oracle import oracle as ora
    p0, _, v0 = ora.gen_columns(SEED, 0, 1_000_000, 1 << 20, False)
    m = idx < 1_000_000
    assert np.array_equal(res.column(1)[m], p0[idx[m]]) and np.array_equal(c1[m], v0[idx[m]])
    res.free(); comp.free()
    for ptr in cols:
        eng.free(ptr)


def test_sort_hundred_million_rows(eng):
    n = 100_000_000
    k = eng.alloc(n * 4)
    eng.gen_columns(SEED, 0, n, 1 << 16, True, None, k, None)
    rid = eng.alloc(n * 4)
    eng.upload(rid, np.arange(n, dtype=np.int32))
    t = eng.table_from_device(n, [k, rid], [np.int32, np.int32])
    res = eng.sort(t, 0, [0, 1])
    ks, rs = res.column(0), res.column(1)
    dk = np.diff(ks)
    assert np.all(dk >= 0)                                                 # sorted
    assert np.all(np.diff(rs)[dk == 0] > 0)                                # stable: row ids ascend inside a key
    assert int(rs.astype(np.int64).sum()) == n * (n - 1) // 2              # a permutation (checksum)
    res.free(); eng.free(k); eng.free(rid)


def test_sort_large_geometry_exact(eng):
    """Just above the size at which the radix passes switch to the large tile geometry (2^25 keys, k_sort.hip: one
    1024-thread workgroup per CU, next tile's keys prefetched), a ragged last tile in every slice, all four passes:
    the permutation must be numpy's stable argsort exactly."""
    n = (1 << 25) + 12345
    rng = np.random.default_rng(25)
    keys = rng.integers(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32)
    keys[rng.integers(0, n, size=n // 8)] = keys[0]                        # a long run of equal keys: stability matters
    k, rid = eng.alloc(n * 4), eng.alloc(n * 4)
    eng.upload(k, keys); eng.upload(rid, np.arange(n, dtype=np.uint32))
    t = eng.table_from_device(n, [k, rid], [np.uint32, np.uint32])
    res = eng.sort(t, 0, [0, 1])
    want = np.argsort(keys, kind="stable")
    assert np.array_equal(res.column(1), want.astype(np.uint32))
    assert np.array_equal(res.column(0), keys[want])
    res.free(); eng.free(k); eng.free(rid)


def test_sort_i64_above_2_24_rows_exact(eng):
    """i64 keys spread over 64 bits, more than 2^24 of them: the tuple passes sort by ALL 32 bits of the high word (four
    passes; up to 2^24 keys three passes over the top 24 bits do), runs of equal high words are fixed up; equal keys keep
    their order.  The permutation must be torch's stable argsort exactly."""
    import torch
    from harkdb_amd.dist import tensor_from_ptr
    dev = torch.device("cuda", 0)
    n = (1 << 24) + 54321
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    keys = torch.randint(-2**62, 2**62, (n,), dtype=torch.int64, device=dev, generator=g) * 2 + torch.randint(0, 2, (n,), dtype=torch.int64, device=dev, generator=g)
    dup = torch.randint(0, n, (n // 16,), dtype=torch.int64, device=dev, generator=g)
    keys[dup] = keys[(dup * 7919) % n]                                     # equal keys: stability matters
    keys[:5000] = (keys[:5000] & ~0xFFFFFFFF) | (keys[0] & 0xFFFFFFFF0000) # a few keys that share a high word's prefix
    rid = torch.arange(n, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    t = eng.table_from_device(n, [keys.data_ptr(), rid.data_ptr()], [np.int64, np.int32], keepalive=(keys, rid))
    res = eng.sort(t, 0, [0, 1])
    want = torch.sort(keys, stable=True)
    got_k = tensor_from_ptr(res.device_ptr(0), n, np.int64, dev)
    got_r = tensor_from_ptr(res.device_ptr(1), n, np.int32, dev)
    assert bool((got_k == want.values).all()) and bool((got_r.long() == want.indices).all())
    res.free(); t.free()


def test_shard_larger_than_2_31_rows_is_fed_in_pieces(eng):
    """2^31 + 4100 rows in one shard (25.8 GB of columns): FgbPlan.run splits the call, the
    accumulators merge the pieces; complement counts add up to the row count."""
    from harkdb_amd.engine import FgbPlan
    n, G = (1 << 31) + 4100, 1 << 20
    p, k, v = eng.alloc(n * 4), eng.alloc(n * 4), eng.alloc(n * 4)
    for lo in range(0, n, 1 << 30):                      # the generator entry also takes < 2^32 rows per call
        m = min(1 << 30, n - lo)
        eng.gen_columns(SEED, lo, m, G, True, p + 4 * lo, k + 4 * lo, v + 4 * lo)
    s, c = eng.alloc(G * 4), eng.alloc(G * 8)
    plan = FgbPlan(eng, 1 << 31, G)
    tot = []
    for cmp in (">", "<="):
        plan.reset()
        plan.run(p, cmp, 0.5, k, v, n)
        plan.finish(s, c)
        tot.append(eng.download(c, G, np.int64))
    assert int((tot[0] + tot[1]).sum()) == n and (tot[0] + tot[1]).min() > 0
    assert abs(int(tot[0].sum()) / n - 0.5) < 1e-3
    plan.free()
    for ptr in (p, k, v, s, c):
        eng.free(ptr)


def test_c4_join_share_i64_full_size(eng):
    """configs[3], one GPU's share: 1.25e8 probe rows x 1.25e7 UNIQUE i64 build keys (half of the probe rows find a
    partner).  Checked on the device: pair count == hits, key equality on every output row, reference order (key,
    left row, right row), every matching probe row present exactly once; plus a 2e6-row probe prefix against the
    numpy model of join.fut:55-75 that tests/test_gpu_groupby_join.py pins to the oracle."""
    import torch
    from harkdb_amd.dist import tensor_from_ptr
    dev = torch.device("cuda", 0)
    n, s = 125_000_000, 12_500_000
    mul = -7046029254386353131                                            # odd: i -> i * mul is a bijection mod 2^64
    bk = torch.arange(s, dtype=torch.int64, device=dev) * mul
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    j = torch.randint(0, 2 * s, (n,), dtype=torch.int64, device=dev, generator=g)
    pk = j * mul
    hits = int((j < s).sum().item())
    prow, brow = torch.arange(n, dtype=torch.int32, device=dev), torch.arange(s, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    tp = eng.table_from_device(n, [pk.data_ptr(), prow.data_ptr()], [np.int64, np.int32], keepalive=(pk, prow))
    tb = eng.table_from_device(s, [bk.data_ptr(), brow.data_ptr()], [np.int64, np.int32], keepalive=(bk, brow))
    res = eng.join(tp, tb, 0, 0, [0, 1], [0, 1])
    P = res.shape[0]
    assert P == hits
    lk, lr, rk, rr = (tensor_from_ptr(res.device_ptr(c), P, dt, dev) for c, dt in ((0, np.int64), (1, np.int32), (2, np.int64), (3, np.int32)))
    assert bool((lk == rk).all())                                          # key equality on every output row
    assert bool((pk[lr.long()] == lk).all()) and bool((bk[rr.long()] == rk).all())    # the row ids point at those keys
    dk = lk[1:] - lk[:-1]
    assert bool((lk[1:] >= lk[:-1]).all())                                 # ascending signed key ...
    assert bool((lr[1:][dk == 0] > lr[:-1][dk == 0]).all())                # ... then left row (unique build keys: one right row per key)
    seen = torch.zeros(n, dtype=torch.bool, device=dev)
    seen[lr.long()] = True
    assert bool((seen == (j < s)).all())                                   # exactly the matching probe rows, each once (P == hits)
    res.free()
    # a probe prefix against the numpy model (build side in full)
    m = 2_000_000
    t2 = eng.table_from_device(m, [pk.data_ptr(), prow.data_ptr()], [np.int64, np.int32], keepalive=(pk, prow))
    r2 = eng.join(t2, tb, 0, 0, [1], [1])
    hk, hb = pk[:m].cpu().numpy(), bk.cpu().numpy()
    ol, orr = np.argsort(hk, kind="stable"), np.argsort(hb, kind="stable")
    sl, sr = hk[ol], hb[orr]
    pos = np.searchsorted(sr, sl)
    ok = (pos < s) & (sr[np.minimum(pos, s - 1)] == sl)
    assert np.array_equal(r2.column(0), ol[ok].astype(np.int32)) and np.array_equal(r2.column(1), orr[pos[ok]].astype(np.int32))
    r2.free()
    # near misses: probe keys one above a build key share its truncated 32-bit word in the bucket kernel's single round
    # (24.4 K keys per bucket here) and must be turned away by the check against the full key
    idx = torch.randint(0, s, (m,), dtype=torch.int64, device=dev, generator=g)
    near = bk[idx] + (torch.arange(m, device=dev) & 1)
    t3 = eng.table_from_device(m, [near.data_ptr(), prow.data_ptr()], [np.int64, np.int32], keepalive=(near, prow))
    r3 = eng.join(t3, tb, 0, 0, [1], [1])
    hk = near.cpu().numpy()
    ol = np.argsort(hk, kind="stable")
    sl = hk[ol]
    pos = np.searchsorted(sr, sl)
    ok = (pos < s) & (sr[np.minimum(pos, s - 1)] == sl)
    assert 0.45 * m < ok.sum() < 0.55 * m
    assert np.array_equal(r3.column(0), ol[ok].astype(np.int32)) and np.array_equal(r3.column(1), orr[pos[ok]].astype(np.int32))
    r3.free()
    for t in (tp, tb, t2, t3):
        t.free()


def test_join_build_side_above_2_24_keys(eng):
    """More than 2^24 unique i64 build keys: the build side takes four tuple passes and a bucket holds more keys than two
    chunks (the truncated-key round does not apply: several rounds over full keys).  Same device-side checks as the
    configs[3] share: pair count, key equality, reference order, every matching probe row exactly once."""
    import torch
    from harkdb_amd.dist import tensor_from_ptr
    dev = torch.device("cuda", 0)
    n, s = 40_000_000, (1 << 24) + 4321
    mul = -7046029254386353131
    bk = torch.arange(s, dtype=torch.int64, device=dev) * mul
    g = torch.Generator(device=dev)
    g.manual_seed(13)
    j = torch.randint(0, 2 * s, (n,), dtype=torch.int64, device=dev, generator=g)
    pk = j * mul
    hits = int((j < s).sum().item())
    prow, brow = torch.arange(n, dtype=torch.int32, device=dev), torch.arange(s, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    tp = eng.table_from_device(n, [pk.data_ptr(), prow.data_ptr()], [np.int64, np.int32], keepalive=(pk, prow))
    tb = eng.table_from_device(s, [bk.data_ptr(), brow.data_ptr()], [np.int64, np.int32], keepalive=(bk, brow))
    res = eng.join(tp, tb, 0, 0, [0, 1], [1])
    P = res.shape[0]
    assert P == hits
    lk, lr, rr = (tensor_from_ptr(res.device_ptr(c), P, dt, dev) for c, dt in ((0, np.int64), (1, np.int32), (2, np.int32)))
    assert bool((pk[lr.long()] == lk).all()) and bool((bk[rr.long()] == lk).all())     # the row ids point at the pair's key
    dk = lk[1:] - lk[:-1]
    assert bool((lk[1:] >= lk[:-1]).all()) and bool((lr[1:][dk == 0] > lr[:-1][dk == 0]).all())   # (key, left row) order
    seen = torch.zeros(n, dtype=torch.bool, device=dev)
    seen[lr.long()] = True
    assert bool((seen == (j < s)).all())
    res.free(); tp.free(); tb.free()


@pytest.mark.parametrize("distinct", [1 << 21, 1 << 31])
def test_reference_groupby_sparse_keys_at_size(eng, distinct):
    """query_groupby (sum, max, min of one column) over 3e7 rows of sparse u32 keys: 2^21 distinct keys take the hash path
    (one partition for the three aggregates), ~3e7 distinct ones the sort path (radix sort + the two-pass segmented tail).
    Checked on the device: the keys are torch.unique's, and every aggregate against a scatter-reduce of the same rows."""
    import torch
    from harkdb_amd.dist import tensor_from_ptr
    dev = torch.device("cuda", 0)
    n = 30_000_000
    g = torch.Generator(device=dev)
    g.manual_seed(17)
    k64 = (torch.randint(0, distinct, (n,), dtype=torch.int64, device=dev, generator=g) * 2654435761) % (1 << 32)
    v64 = torch.randint(0, 1 << 32, (n,), dtype=torch.int64, device=dev, generator=g)
    k32 = (k64 - ((k64 >> 31) << 32)).to(torch.int32)                       # the u32 bit patterns as int32 tensors
    v32 = (v64 - ((v64 >> 31) << 32)).to(torch.int32)
    torch.cuda.synchronize()
    t = eng.table_from_device(n, [k32.data_ptr(), v32.data_ptr()], [np.uint32, np.uint32], keepalive=(k32, v32))
    res = eng.query_groupby(t, 0, [1, 1, 1], [2, 3, 4])
    uk, inv = torch.unique(k64, sorted=True, return_inverse=True)
    G = uk.numel()
    assert res.shape == (G, 4)
    got = [tensor_from_ptr(res.device_ptr(c), G, np.int32, dev).long() & 0xFFFFFFFF for c in range(4)]
    assert bool((got[0] == uk).all())                                        # ascending unsigned keys
    ssum = torch.zeros(G, dtype=torch.int64, device=dev).scatter_add_(0, inv, v64) & 0xFFFFFFFF     # + mod 2^32 (groupby.fut:37)
    smax = torch.zeros(G, dtype=torch.int64, device=dev).scatter_reduce_(0, inv, v64, "amax")
    smin = torch.full((G,), 1 << 32, dtype=torch.int64, device=dev).scatter_reduce_(0, inv, v64, "amin")
    assert bool((got[1] == ssum).all()) and bool((got[2] == smax).all()) and bool((got[3] == smin).all())
    res.free(); t.free()


def test_c5_full_pipeline_share(eng):
    """configs[4], one GPU's share: 5e8 rows x (i32 key + 16 f32 columns) = 34 GB resident, the full SELECT / WHERE /
    GROUP BY / HAVING / ORDER BY / LIMIT statement through the SQL surface.  Properties: linearity in the predicate,
    HAVING complement, ORDER BY sortedness, LIMIT prefix == prefix of the unlimited result."""
    from harkdb_amd import FutharkContext
    n, G = 500_000_000, 1 << 20
    fc = FutharkContext.__new__(FutharkContext)
    fc.FutEnv, fc.tables, fc.sql_mode = eng, {}, True
    cols = [eng.alloc(n * 4) for _ in range(16)]
    key = eng.alloc(n * 4)
    for j in range(0, 16, 2):
        eng.gen_columns(SEED + j, 0, n, G, False, cols[j], key if j == 0 else None, cols[j + 1])
    fc.create_table_from_device("t", ["k"] + [f"c{j}" for j in range(16)], [key] + cols, [np.int32] + [np.float32] * 16, n)
    q = lambda s: fc.sql_columns(s)[1]
    k_gt, s_gt, c_gt = q("select k, sum(c3), count(*) from t where c1 > 0.5 group by k")
    k_le, s_le, c_le = q("select k, sum(c3), count(*) from t where c1 <= 0.5 group by k")
    k_all, s_all, c_all = q("select k, sum(c3), count(*) from t group by k")
    assert len(k_all) == G and np.array_equal(k_all, np.arange(G)) and np.array_equal(k_gt, k_all) and np.array_equal(k_le, k_all)
    assert int(c_all.sum()) == n and np.array_equal(c_gt + c_le, c_all)                      # counts: exact linearity
    assert np.allclose(s_gt.astype(np.float64) + s_le.astype(np.float64), s_all.astype(np.float64), rtol=1e-5)   # f32 sums: 1e-5 relative
    assert abs(int(c_gt.sum()) / n - 0.5) < 1e-3
    # HAVING complement over the filtered aggregation
    kh, ch = q("select k, count(*) from t where c1 > 0.5 group by k having count(*) > 250")
    kl, cl = q("select k, count(*) from t where c1 > 0.5 group by k having count(*) <= 250")
    assert len(kh) + len(kl) == G and ch.min() > 250 and (len(cl) == 0 or cl.max() <= 250)
    assert np.array_equal(np.sort(np.concatenate([kh, kl])), k_all) and np.array_equal(ch, c_gt[kh])
    # the full statement: ORDER BY sortedness, LIMIT prefix
    stmt = "select k, sum(c3), count(*), avg(c3) from t where c1 > 0.5 group by k having count(*) > 250 order by sum(c3) desc"
    ko, so, co, ao = q(stmt)
    assert len(ko) == len(kh) and np.all(np.diff(so) <= 0)                                    # sorted descending by the aggregate
    assert np.array_equal(so, s_gt[ko]) and np.array_equal(co, c_gt[ko])                      # the rows are the aggregation's rows
    assert np.allclose(ao, so.astype(np.float64) / co, rtol=1e-6)
    kl10, sl10, cl10, al10 = q(stmt + " limit 10")
    assert np.array_equal(kl10, ko[:10]) and np.array_equal(sl10, so[:10]) and np.array_equal(cl10, co[:10]) and np.array_equal(al10, ao[:10])
    # three aggregates over three columns + MAX/MIN bounds from the generator (values uniform in [0, 1))
    k3, s3, mx, mn, c3 = q("select k, sum(c3), max(c7), min(c9), count(*) from t where c1 > 0.5 group by k")
    assert np.array_equal(s3, s_gt) and np.array_equal(c3, c_gt) and mx.max() < 1.0 and mn.min() >= 0.0 and np.all(mx[c3 > 50] > mn[c3 > 50])
    # ... which took the pair pass (sum(c3) + max(c7) in one producer + consumer pass): the single passes give the same bits
    import os
    os.environ["HARK_NO_PAIR_PASS"] = "1"
    try:
        k1, s1, mx1, mn1, c1 = q("select k, sum(c3), max(c7), min(c9), count(*) from t where c1 > 0.5 group by k")
    finally:
        del os.environ["HARK_NO_PAIR_PASS"]
    assert np.array_equal(k1, k3) and np.array_equal(s1, s3) and np.array_equal(mx1, mx) and np.array_equal(mn1, mn) and np.array_equal(c1, c3)
    # the statement above was ONE pass over the rows (triple pass: 14-byte entries in slabs sized for them, the consumer's
    # 20 B x 8192 keys fill a CU's 160 KiB of LDS), also when 90 % of the rows survive; with the triple pass switched off a
    # pair pass + a single pass give the same rows
    fc.sql("select k, sum(c3), max(c7), min(c9), count(*) from t where c1 > 0.5 group by k")
    assert fc.FutEnv.last_groupby_passes() == 1
    many = q("select k, sum(c3), max(c7), min(c9), count(*) from t where c1 > 0.1 group by k")
    assert fc.FutEnv.last_groupby_passes() == 1
    os.environ["HARK_NO_TRIPLE_PASS"] = "1"
    try:
        many2 = q("select k, sum(c3), max(c7), min(c9), count(*) from t where c1 > 0.1 group by k")
        assert fc.FutEnv.last_groupby_passes() == 2
        k2, s2, mx2, mn2, c2 = q("select k, sum(c3), max(c7), min(c9), count(*) from t where c1 > 0.5 group by k")
    finally:
        del os.environ["HARK_NO_TRIPLE_PASS"]
    assert all(np.array_equal(x, y) for x, y in zip(many, many2))
    assert np.array_equal(k2, k3) and np.array_equal(s2, s3) and np.array_equal(mx2, mx) and np.array_equal(mn2, mn) and np.array_equal(c2, c3)
    # late aggregation: with LIMIT 10, max(c7) and min(c9) are computed for the ten surviving groups only (one pass over
    # c1 and k, hark_entry_filter_groupby_subset): the rows are the first ten of the unlimited ordered statement, bit for bit
    stmt3 = "select k, sum(c3), max(c7), min(c9), count(*) from t where c1 > 0.5 group by k having count(*) > 250 order by sum(c3) desc"
    ka, sa, mxa, mna, ca = q(stmt3)
    kb, sb, mxb, mnb, cb = q(stmt3 + " limit 10")
    assert len(kb) == 10 and np.array_equal(kb, ka[:10]) and np.array_equal(sb, sa[:10]) and np.array_equal(cb, ca[:10])
    assert np.array_equal(mxb, mxa[:10]) and np.array_equal(mnb, mna[:10]) and np.array_equal(mxb, mx[kb]) and np.array_equal(mnb, mn[kb])
    fc.drop_table("t")
    for ptr in cols + [key]:
        eng.free(ptr)


def test_c3_one_billion_rows_sorted_by_the_key_three_paths_agree(eng):
    """configs[2] + filter on the 1e9-row table SORTED by k (a table kept in key order): the window path (what the test kernel picks),
    the plain partition (window = 2) and the partition with rotated loads (window = 3) give the same sums and counts per group, bit
    for bit (integer-valued f32 values), with the predicate, with its complement and without one; linearity holds on the window path."""
    from harkdb_amd.engine import FgbPlan
    n, G = 1_000_000_000, 1 << 20
    p, k, v = eng.alloc(n * 4), eng.alloc(n * 4), eng.alloc(n * 4)
    eng.gen_columns(SEED, 0, n, G, True, p, k, v)
    tk = eng.table_from_device(n, [k], [np.int32])
    srt = eng.sort(tk, 0, [0])                                          # the key column in ascending order (p and v keep their rows)
    ks = srt.device_ptr(0)
    s, c = eng.alloc(G * 4), eng.alloc(G * 8)
    got = {}
    for path, knobs in (("window", {}), ("partition", {"window": 2}), ("rotated", {"window": 3})):
        plan = FgbPlan(eng, n, G, timing=1, **knobs)
        for name, pred, cmp in (("gt", p, ">"), ("le", p, "<="), ("all", None, ">")):
            if path != "window" and name == "le": continue
            plan.reset()
            plan.run(pred, cmp, 0.5, ks, v, n)
            plan.finish(s, c)
            got[path, name] = (eng.download(s, G, np.float32), eng.download(c, G, np.int64))
        _, launches = plan.timing()
        assert (launches["consumer"] == 0) == (path == "window"), (path, launches)     # the window path has no consumer pass
        plan.free()
    for name in ("gt", "all"):
        for path in ("partition", "rotated"):
            assert np.array_equal(got["window", name][1], got[path, name][1]) and np.array_equal(got["window", name][0], got[path, name][0]), (path, name)
    assert np.array_equal(got["window", "gt"][1] + got["window", "le"][1], got["window", "all"][1])
    assert np.array_equal(got["window", "gt"][0].astype(np.float64) + got["window", "le"][0].astype(np.float64), got["window", "all"][0].astype(np.float64))
    assert got["window", "all"][1].sum() == n
    srt.free(); tk.free()
    for ptr in (p, k, v, s, c):
        eng.free(ptr)


@pytest.mark.parametrize("order", ["sorted", "descending", "runs"])
def test_join_one_hundred_million_probe_rows_in_key_order(eng, order):
    """1e8 probe rows x 1e7 build rows (u32 keys, duplicates on the build side), the PROBE column sorted by the key, descending, or in
    sorted runs of 4096 rows in shuffled order: the search path (k_cjoin.hip) -- pair count, key equality, the reference's order (key,
    left row, right row) and every matching probe row with all its partners, checked on the device; the same rows as the partitioned
    path gives for the shuffled column."""
    import torch
    from harkdb_amd.dist import tensor_from_ptr
    dev = torch.device("cuda", 0)
    n, s = 100_000_000, 10_000_000
    g = torch.Generator(device=dev)
    g.manual_seed(29)
    bk = torch.randint(0, 1 << 27, (s,), dtype=torch.int32, device=dev, generator=g)        # ~7 % of the build keys occur twice or more
    pk = torch.randint(0, 1 << 28, (n,), dtype=torch.int32, device=dev, generator=g)        # half of the probe keys lie outside the build side's range
    pk = pk.sort(descending=(order == "descending")).values
    if order == "runs":
        pk = pk[: n // 4096 * 4096].view(-1, 4096)[torch.randperm(n // 4096, device=dev, generator=g)].reshape(-1).contiguous()
        n = pk.numel()
    shuffled = pk[torch.randperm(n, device=dev, generator=g)].contiguous()                  # the same column, its rows shuffled
    prow, brow = torch.arange(n, dtype=torch.int32, device=dev), torch.arange(s, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    tb = eng.table_from_device(s, [bk.data_ptr(), brow.data_ptr()], [np.uint32, np.int32], keepalive=(bk, brow))
    tp = eng.table_from_device(n, [pk.data_ptr(), prow.data_ptr()], [np.uint32, np.int32], keepalive=(pk, prow))
    res = eng.join(tp, tb, 0, 0, [0, 1], [1])
    assert eng.last_join_path().startswith("clustered"), eng.last_join_path()
    P = res.shape[0]
    lk, lr, rr = (tensor_from_ptr(res.device_ptr(c), P, dt, dev) for c, dt in ((0, np.int32), (1, np.int32), (2, np.int32)))
    assert bool((pk[lr.long()] == lk).all()) and bool((bk[rr.long()] == lk).all())           # the row ids point at the pair's key
    same = lk[1:] == lk[:-1]
    assert bool((lk[1:] >= lk[:-1]).all())                                                   # keys ascend (all below 2^31: signed = unsigned)
    assert bool((lr[1:][same] >= lr[:-1][same]).all())                                       # ... then left rows
    same2 = same & (lr[1:] == lr[:-1])
    assert bool((rr[1:][same2] > rr[:-1][same2]).all())                                      # ... then right rows
    # every probe row with all its partners: pairs per probe row = occurrences of its key on the build side
    occ = torch.zeros(1 << 27, dtype=torch.int32, device=dev)
    occ.index_add_(0, bk.long(), torch.ones(s, dtype=torch.int32, device=dev))
    want = torch.where(pk < (1 << 27), occ[pk.clamp(max=(1 << 27) - 1).long()], torch.zeros((), dtype=torch.int32, device=dev))
    have = torch.zeros(n, dtype=torch.int32, device=dev)
    have.index_add_(0, lr.long(), torch.ones(P, dtype=torch.int32, device=dev))
    assert P == int(want.sum().item()) and bool((have == want).all())
    checksum = int((lk.long() * 31 + rr.long()).sum().item())
    res.free(); tp.free()
    # the shuffled column through the partitioned path: the same multiset of (key, right row) pairs
    ts = eng.table_from_device(n, [shuffled.data_ptr(), prow.data_ptr()], [np.uint32, np.int32], keepalive=(shuffled, prow))
    res2 = eng.join(ts, tb, 0, 0, [0, 1], [1])
    assert eng.last_join_path().startswith("partitioned") and res2.shape[0] == P
    k2, r2 = tensor_from_ptr(res2.device_ptr(0), P, np.int32, dev), tensor_from_ptr(res2.device_ptr(2), P, np.int32, dev)
    assert int((k2.long() * 31 + r2.long()).sum().item()) == checksum
    res2.free(); ts.free(); tb.free()
