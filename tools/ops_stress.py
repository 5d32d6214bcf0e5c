"""Randomized differential stress of sort / join / group-by / filter entries against numpy models (sizes straddle the
path thresholds: 2^18 rows for the hash group-by, 2^20 rows and 4x for the join pre-filter).
Usage: python tools/ops_stress.py [seconds] [seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from harkdb_amd.engine import Engine

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
eng = Engine(0)
M32 = np.uint64(0xFFFFFFFF)


def rand_keys(n, dtype, nd):
    if dtype == np.float32:
        pool = (rng.standard_normal(nd) * 100).astype(np.float32)
    elif dtype == np.int64:
        pool = rng.integers(-2**62, 2**62, size=nd)
    elif dtype == np.int32:
        pool = rng.integers(-2**31, 2**31, size=nd).astype(np.int32)
    else:
        pool = rng.integers(0, 2**32, size=nd, dtype=np.uint64).astype(np.uint32)
    if rng.random() < 0.3:                                   # narrow range: radix passes get skipped
        pool = (pool.astype(np.int64) % 1000).astype(dtype) if dtype != np.float32 else (pool % 8).astype(np.float32)
    return pool[rng.integers(0, nd, size=n)]


def join_rows(lk, rk):
    ol, orr = np.argsort(lk, kind="stable"), np.argsort(rk, kind="stable")
    sl, sr = lk[ol], rk[orr]
    lb, ub = np.searchsorted(sr, sl, "left"), np.searchsorted(sr, sl, "right")
    cnt = ub - lb
    li = np.repeat(ol, cnt)
    within = np.arange(int(cnt.sum())) - np.repeat(np.cumsum(cnt) - cnt, cnt)
    return li, orr[np.repeat(lb, cnt) + within]


def case_sort():
    n = int(rng.choice([1, 100, 4097, 70_000, 600_000]))
    dt = rng.choice([np.uint32, np.int32, np.float32, np.int64])
    key = rand_keys(n, dt, int(rng.integers(1, max(2, n))))
    a, b = rng.integers(-9, 9, n).astype(np.int32), rng.random(n).astype(np.float32)
    cols = [int(c) for c in rng.integers(0, 3, size=int(rng.integers(1, 5)))]
    desc = bool(rng.random() < 0.5)
    t = eng.table_from_columns([key, a, b])
    res = eng.sort(t, 0, cols, descending=desc)
    order = key.astype(np.float64) if dt != np.int64 else key
    perm = np.argsort(-order if desc else order, kind="stable") if dt != np.int64 else np.argsort(-key if desc else key, kind="stable")
    src = [key, a, b]
    ok = all(np.array_equal(res.column(j).view(np.uint8), src[c][perm].view(np.uint8)) for j, c in enumerate(cols))
    res.free(); t.free()
    return ok, dict(op="sort", n=n, dt=dt.__name__, cols=cols, desc=desc)


def case_join():
    n = int(rng.choice([50, 5000, (1 << 20) - 7, (1 << 20) + 9, 1_500_000]))
    s = int(rng.choice([1, 300, n // 5 + 1, n // 3 + 1, 200_000]))
    wide = bool(rng.random() < 0.4)
    dt = np.int64 if wide else np.uint32
    nd = int(rng.integers(1, 400_000))
    pool = rand_keys(nd, dt, nd)
    lk, rk = pool[rng.integers(0, nd, n)], pool[rng.integers(0, nd, s)]
    if rng.random() < 0.5:                                   # few hits: most probe keys are foreign
        miss = rng.random(n) < 0.9
        lk = np.where(miss, rand_keys(n, dt, n), lk)
    la, rb = np.arange(n, dtype=np.int32), rng.integers(0, 99, s).astype(np.int32)
    ul, cl = np.unique(lk, return_counts=True)
    ur, cr = np.unique(rk, return_counts=True)
    _, il, ir = np.intersect1d(ul, ur, return_indices=True)
    if int((cl[il].astype(np.int64) * cr[ir]).sum()) > 30_000_000:          # a near cross product: not a useful case
        return True, dict(op="join-skip")
    t1, t2 = eng.table_from_columns([lk, la]), eng.table_from_columns([rb, rk])
    # which columns: only the carried / rank-ordered ones (no row ids, no permutation needed), or keys and repeats as well
    c1, c2 = [([1], [0]), ([0, 1], [0, 1]), ([1, 1, 0], [1, 0, 0])][int(rng.integers(0, 3))]
    res = eng.join(t1, t2, 0, 1, c1, c2)
    cmp_l, cmp_r = (lk, rk) if wide else (lk.astype(np.uint32), rk.astype(np.uint32))
    li, ri = join_rows(cmp_l, cmp_r)
    exp = [(lk, la)[c][li] for c in c1] + [(rb, rk)[c][ri] for c in c2]
    ok = res.shape[0] == len(li) and (len(li) == 0 or all(np.array_equal(res.column(j), e) for j, e in enumerate(exp)))
    res.free(); t1.free(); t2.free()
    return ok, dict(op="join", n=n, s=s, wide=wide, nd=nd)


def case_groupby():
    n = int(rng.choice([10, 3000, (1 << 18) - 3, (1 << 18) + 5, 900_000]))
    kind = rng.choice(["dense", "sparse", "few"])
    if kind == "dense":
        G = int(rng.choice([5, 9000, 1 << 20]))
        k = rng.integers(0, G, n).astype(np.uint32)
    elif kind == "sparse":
        nd = int(rng.integers(1, max(2, n)))
        k = rand_keys(n, np.uint32, nd)
    else:
        k = rng.choice(rng.integers(0, 2**32, size=3, dtype=np.uint64).astype(np.uint32), size=n)
    v1 = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    v2 = rng.integers(0, 7, n).astype(np.uint32) * 2 + 1
    v3 = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    v4 = rng.integers(0, 1000, n).astype(np.uint32)
    s_cols = [int(c) for c in rng.integers(1, 5, size=int(rng.integers(1, 7)))]     # up to six aggregates of four columns: statistics, triple, pair and
    t_cols = [int(c) for c in rng.integers(1, 5, size=len(s_cols))]                 # multi-operator hash passes
    t = eng.table_from_columns([k, v1, v2, v3, v4])
    res = eng.query_groupby(t, 0, s_cols, t_cols)
    got = [res.column(j) for j in range(1 + len(s_cols))]
    order = np.argsort(k, kind="stable")
    ks = k[order]
    heads = np.flatnonzero(np.r_[True, ks[1:] != ks[:-1]])
    ok = np.array_equal(got[0].view(np.uint32), ks[heads])
    for j, (c, op) in enumerate(zip(s_cols, t_cols)):
        v = [None, v1, v2, v3, v4][c][order].astype(np.uint64)
        if op == 2:
            e = np.add.reduceat(v, heads) & M32
        elif op == 3:
            e = np.maximum.reduceat(v, heads)
        elif op == 4:
            e = np.minimum.reduceat(v, heads)
        else:                                                # wrapping product: fold segment by segment in 32-bit halves
            e = np.empty(len(heads), dtype=np.uint64)
            ends = np.r_[heads[1:], len(v)]
            if len(heads) > 20000:                           # keep the python loop short
                res.free(); t.free()
                return True, dict(op="groupby-skip")
            for g, (a, b) in enumerate(zip(heads, ends)):
                p = 1
                for x in v[a:b].tolist():
                    p = (p * x) & 0xFFFFFFFF
                e[g] = p
        ok = ok and np.array_equal(got[1 + j].view(np.uint32).astype(np.uint64), e)
    res.free(); t.free()
    return ok, dict(op="groupby", n=n, kind=str(kind), s_cols=s_cols, t_cols=t_cols)


def case_filter():
    n = int(rng.choice([1, 4095, 4097, 300_000, 2_000_000]))
    dt = rng.choice([np.float32, np.int32, np.uint32, np.int64])
    col = rand_keys(n, dt, int(rng.integers(1, max(2, n))))
    a = rng.integers(0, 99, n).astype(np.int32)
    cmp = str(rng.choice([">", ">=", "<", "<=", "=", "!="]))
    c = col[int(rng.integers(0, n))]
    t = eng.table_from_columns([col, a])
    res = eng.filter_sel(t, 0, cmp, c.item(), [1, 0], want_row_index=True)
    keep = {">": col > c, ">=": col >= c, "<": col < c, "<=": col <= c, "=": col == c, "!=": col != c}[cmp]
    idx = np.flatnonzero(keep)
    ok = res.shape[0] == len(idx) and np.array_equal(res.column(0), idx) and np.array_equal(res.column(1), a[idx]) and \
        np.array_equal(res.column(2).view(np.uint8), col[idx].view(np.uint8))
    res.free(); t.free()
    return ok, dict(op="filter", n=n, dt=dt.__name__, cmp=cmp)


cases, t_end, bad = 0, time.time() + budget, None
counts = {}
t_note = time.time() + 60
while time.time() < t_end and bad is None:
    if time.time() > t_note:                                  # a sign of life per minute (a silent GPU job is taken to be hung)
        print(f"... {cases} cases so far", flush=True); t_note = time.time() + 60
    fn = [case_sort, case_join, case_groupby, case_filter][int(rng.integers(0, 4))]
    ok, info = fn()
    cases += 1
    counts[info["op"]] = counts.get(info["op"], 0) + 1
    if not ok:
        bad = info
print(f"{cases} random cases {counts}: {'MISMATCH ' + str(bad) if bad else 'all exact'}", flush=True)
sys.exit(1 if bad else 0)
