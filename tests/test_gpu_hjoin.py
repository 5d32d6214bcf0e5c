"""The partitioned join (harkdb_amd/csrc/k_hjoin.hip: probe side range-partitioned by splitters of the sorted build
side, build slices staged in LDS) against the reference's order -- ascending key, left row, right row
(join.fut:55-75) -- computed by an independent numpy model that tests/test_gpu_groupby_join.py pins to the oracle.
It runs for probe sides >= 2^18 rows with >= 4096 build rows; everything else takes the sort-merge path."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _np_join_rows(lk, rk):
    ol, orr = np.argsort(lk, kind="stable"), np.argsort(rk, kind="stable")
    sl, sr = lk[ol], rk[orr]
    lb, ub = np.searchsorted(sr, sl, "left"), np.searchsorted(sr, sl, "right")
    cnt = ub - lb
    li = np.repeat(ol, cnt)
    within = np.arange(int(cnt.sum())) - np.repeat(np.cumsum(cnt) - cnt, cnt)
    ri = orr[np.repeat(lb, cnt) + within]
    return li, ri


@pytest.fixture(scope="module")
def eng():
    from harkdb_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def _check(eng, lk, rk):
    n, s = len(lk), len(rk)
    la, rb = np.arange(n, dtype=np.int32), np.arange(s, dtype=np.int32)
    t1, t2 = eng.table_from_columns([lk, la]), eng.table_from_columns([rb, rk])
    res = eng.join(t1, t2, 0, 1, [1, 0], [0])
    # 32-bit keys join as u32 bit patterns (join.fut:52 types both tables u32): negative i32 keys sort after the others
    li, ri = _np_join_rows(lk.view(np.uint32), rk.view(np.uint32)) if lk.dtype.itemsize == 4 else _np_join_rows(lk, rk)
    assert res.shape[0] == len(li)
    if len(li):
        assert np.array_equal(res.column(0), la[li])          # left row ids: bit-exact order (key, left row, right row)
        assert np.array_equal(res.column(2), rb[ri])
        assert np.array_equal(res.column(1), lk[li])
    res.free()
    t1.free()
    t2.free()
    return len(li)


CASES = {
    "u32 10% hits": dict(dt=np.uint32, n=(1 << 19) + 777, s=50_000, pool=None, hit=0.1),
    "u32 every row hits, 8 partners each": dict(dt=np.uint32, n=300_000, s=40_000, pool=5_000, hit=1.0),
    "u32 few distinct keys (duplicate splitters)": dict(dt=np.uint32, n=270_000, s=8_192, pool=50, hit=0.002),
    "i32 negative keys as u32": dict(dt=np.int32, n=280_000, s=10_000, pool=None, hit=0.3),
    "i64 signed keys": dict(dt=np.int64, n=(1 << 18) + 5, s=30_000, pool=None, hit=0.5),
    "i64 duplicates both sides": dict(dt=np.int64, n=300_001, s=20_000, pool=4_000, hit=0.2),
    "build side larger than probe": dict(dt=np.uint32, n=1 << 18, s=600_000, pool=None, hit=0.4),
    "u32 unique build keys (primary key): no counts, no expansion": dict(dt=np.uint32, n=400_000, s=60_000, pool=None, hit=0.5, unique=True),
    "i64 unique build keys, every second probe row hits": dict(dt=np.int64, n=(1 << 19) + 3, s=100_000, pool=None, hit=0.5, unique=True),
    # the build side's i64 sort orders by the high word first and fixes runs of equal high words in registers (<= 16 keys):
    "i64 build keys in short runs of equal high words (2..16)": dict(dt=np.int64, n=300_000, s=60_000, pool=None, hit=0.4, hiruns=16),
    "i64 build keys with a run of 17 equal high words (eight-pass fallback)": dict(dt=np.int64, n=300_000, s=60_000, pool=None, hit=0.4, hiruns=17),
    "i64 build keys below 2^40 (few distinct high words: fallback)": dict(dt=np.int64, n=280_000, s=50_000, pool=None, hit=0.3, small=True),
    # a bucket of up to two chunks of i64 build keys is probed in ONE round over keys truncated to 32 bits; consecutive keys
    # beside keys spread over 64 bits in one bucket truncate to equal words -> that bucket takes the rounds instead
    "i64 consecutive keys among spread ones": dict(dt=np.int64, n=300_000, s=60_000, pool=None, hit=0.5, clustered=True),
}


def _make(case, seed):
    c = CASES[case]
    rng = np.random.default_rng(seed)
    dt, n, s = c["dt"], c["n"], c["s"]
    info = np.iinfo(dt)
    if c["pool"]:
        pool = rng.integers(info.min, info.max, size=c["pool"], dtype=np.int64).astype(dt)
        rk = pool[rng.integers(0, len(pool), size=s)]
    elif c.get("hiruns"):
        m = c["hiruns"]
        hi = rng.integers(-2**31, 2**31, size=s // 4, dtype=np.int64)
        hi = np.repeat(hi, rng.integers(1, 8, size=len(hi)))[:s]
        hi[:m] = 123456                                                   # one run of exactly m equal high words
        hi = np.resize(hi, s)
        rk = ((hi << 32) | rng.integers(0, 2**32, size=s, dtype=np.int64)).astype(np.int64)
        rk[1] = rk[0]                                                     # and a fully equal pair inside it
        rk = rk[rng.permutation(s)]
    elif c.get("small"):
        rk = rng.integers(-2**39, 2**39, size=s, dtype=np.int64)
    elif c.get("clustered"):
        rk = np.concatenate([np.arange(s // 2, dtype=np.int64) * 3 + 10**12, rng.integers(info.min, info.max, size=s - s // 2, dtype=np.int64)])
        rk[:50] = rk[50:100]                                              # some duplicates inside the cluster
        rk = rk[rng.permutation(s)]
    elif c.get("unique"):
        rk = (rng.permutation(s).astype(np.int64) * 40503 - 1_000_000_007).astype(dt)     # distinct, scattered, some negative
    else:
        rk = rng.integers(info.min, info.max, size=s, dtype=np.int64).astype(dt)
    lk = rng.integers(info.min, info.max, size=n, dtype=np.int64).astype(dt)
    hit = rng.random(n) < c["hit"]
    lk[hit] = rk[rng.integers(0, s, size=int(hit.sum()))]
    return lk, rk


@pytest.mark.parametrize("case", list(CASES))
def test_partitioned_join_reference_order(eng, case):
    lk, rk = _make(case, len(case))
    pairs = _check(eng, lk, rk)
    if CASES[case]["hit"] >= 0.1:
        assert pairs > 0


@pytest.mark.parametrize("unique", [True, False, "spread", "spread_dup"])
def test_i64_join_with_several_output_columns(eng, unique):
    """Output columns of the i64 path arrive in three ways: ONE probe-side column travels with the probe rows (the fourth
    word of their 16-byte entries), ONE build-side column is read off in rank order by the order kernel (unique build
    keys) or through the pair's rank, everything else -- further columns, f32, a column selected twice, the key itself --
    is gathered.  All of them against numpy, row for row."""
    rng = np.random.default_rng(21 + ["False", "True", "spread", "spread_dup"].index(str(unique)))
    n, s = 300_007, 40_000
    if unique == "spread":                 # build keys spread over 64 bits: key, row id and the build-side column travel through
        rk = np.unique(rng.integers(-2**63, 2**63 - 1, size=s + 100))[:s]          # the sort as 16-byte tuples (sort_i64_tuples)
        rk = rk[rng.permutation(s)]
    elif unique == "spread_dup":           # ... with runs of equal keys and of keys that share their 24-bit prefix
        pool = rng.integers(-2**63, 2**63 - 1, size=s // 4)
        rk = pool[rng.integers(0, len(pool), size=s)] + rng.integers(0, 3, size=s) * rng.integers(0, 2, size=s)
    else:
        rk = (rng.permutation(s).astype(np.int64) * 7919 - 10**9) if unique else rng.integers(-5000, 5000, size=s).astype(np.int64)
    lk = rng.integers(-2**62, 2**62, size=n).astype(np.int64)
    hit = rng.random(n) < (0.5 if unique in (True, "spread") else 0.02)
    lk[hit] = rk[rng.integers(0, s, size=int(hit.sum()))]
    la, lf, lc = rng.integers(-2**31, 2**31, n).astype(np.int32), rng.random(n).astype(np.float32), rng.integers(0, 99, n).astype(np.int32)
    ra, rf = rng.integers(0, 2**31, s).astype(np.int32), rng.random(s).astype(np.float32)
    t1, t2 = eng.table_from_columns([lk, la, lf, lc]), eng.table_from_columns([rk, ra, rf])
    res = eng.join(t1, t2, 0, 0, [2, 1, 0, 3, 1], [2, 1, 0, 1])
    li, ri = _np_join_rows(lk, rk)
    assert res.shape == (len(li), 9) and len(li) > 0
    for j, exp in enumerate([lf[li], la[li], lk[li], lc[li], la[li], rf[ri], ra[ri], rk[ri], ra[ri]]):
        assert np.array_equal(res.column(j), exp), j
    res.free(); t1.free(); t2.free()


def test_keys_outside_the_build_range_and_exact_batch_multiples(eng):
    rng = np.random.default_rng(9)
    s, n = 4096, 1 << 18                                         # n is a multiple of the 4096-row batch
    rk = rng.integers(1000, 2000, size=s).astype(np.uint32)
    lk = rng.integers(0, 3000, size=n).astype(np.uint32)         # two thirds of the probe keys lie outside [min, max] of the build side
    assert _check(eng, lk, rk) > 0
    assert _check(eng, (lk + np.uint32(5000)), rk) == 0          # no probe key inside the range at all


def test_one_build_key_repeated_past_the_order_kernels_counters(eng):
    """A build key repeated more than kCoarse * kFine (2048 * 8192) times makes ONE bucket of that many sorted build
    entries (the splitters never cut a run of equal keys): a group of ranks no longer fits the order kernel's fine
    counters, so it must hand the bucket to the general radix path instead of counting past its LDS arrays.  Probe
    rows hit the neighbours of the run (same bucket), never the run itself (that would be a 1.7e7-fold fan-out)."""
    rng = np.random.default_rng(12)
    hot, reps, others, n = np.uint32(0x40000000), (1 << 24) + 5000, 200_000, (1 << 18) + 11
    rest = rng.integers(0, 2**32, size=others, dtype=np.uint64).astype(np.uint32)
    rest = rest[rest != hot]
    near = (hot + 1 + rng.integers(0, 1 << 20, size=3000).astype(np.uint32)).astype(np.uint32)    # keys right behind the run
    rk = np.concatenate([np.full(reps, hot, dtype=np.uint32), rest, near])
    rk = rk[rng.permutation(len(rk))]
    lk = rng.integers(0, 2**32, size=n, dtype=np.uint64).astype(np.uint32)
    lk[lk == hot] += np.uint32(1)
    hit = rng.random(n) < 0.2
    pool = np.concatenate([near, rest[:5000]])
    lk[hit] = pool[rng.integers(0, len(pool), size=int(hit.sum()))]
    assert _check(eng, lk, rk) > 0


def test_every_probe_row_on_one_key_stays_partitioned(eng):
    """Every probe row carries one key: the sample finds it, its rows never enter the partition (they would overflow their
    bucket's slabs) and arrive as one block of the output, in row order, from the stable partition of the probe column."""
    rng = np.random.default_rng(10)
    s, n = 5000, 300_000
    rk = rng.integers(0, 2**32, size=s, dtype=np.uint64).astype(np.uint32)
    lk = np.full(n, rk[17], dtype=np.uint32)
    assert _check(eng, lk, rk) >= n
    assert eng.last_join_path().startswith("partitioned")


def _skewed(rng, dt, n, s, dup, shares, absent=0.0):
    """Probe keys: 20 % spread over the build keys, `shares[i]` of the rows on hot key i (neighbours in the sorted build
    side: one bucket), `absent` of them on a key the build side does not have."""
    info = np.iinfo(dt)
    rk = rng.integers(info.min, info.max, size=s, dtype=np.int64).astype(dt)
    if dup:
        rk[: s // 3] = rk[s // 3: 2 * (s // 3)]                        # a third of the build keys twice
    srt = np.unique(rk)
    hot = srt[len(srt) // 2: len(srt) // 2 + len(shares)]              # consecutive build keys
    lk = rng.integers(info.min, info.max, size=n, dtype=np.int64).astype(dt)
    u = rng.random(n)
    lk[u < 0.2] = rk[rng.integers(0, s, size=int((u < 0.2).sum()))]
    edge = 0.2
    for key, share in zip(hot, shares):
        m = (u >= edge) & (u < edge + share)
        lk[m] = key
        edge += share
    if absent:
        gone = dt(12345)
        assert gone not in set(rk.tolist())
        lk[(u >= edge) & (u < edge + absent)] = gone
    return lk, rk


@pytest.mark.parametrize("dt", [np.uint32, np.int64])
@pytest.mark.parametrize("dup", [False, True])
def test_hot_keys_come_out_of_the_stable_partition(eng, dt, dup):
    """Heavy hitters (k_hjoin.hip, jhot_*): one key with 10 % of the probe rows, thirty with 1 % each -- all in one bucket -- and
    one with 5 % that the build side does not have.  Their rows are blocks of the output between the other rows, in row
    order; the partitioned path must deliver them (no fallback) exactly as the reference orders them (join.fut:55-75)."""
    rng = np.random.default_rng(77 + int(dup) + (3 if dt is np.int64 else 0))
    lk, rk = _skewed(rng, dt, 700_001, 50_000, dup, [0.10] + [0.01] * 30, absent=0.05)
    assert _check(eng, lk, rk) > 300_000
    assert eng.last_join_path().startswith("partitioned")


def test_hot_keys_with_carried_columns_only(eng):
    """The i64 path with only the carried probe-side column and the rank-ordered build-side column selected (the order
    kernel leaves (rank, left row) out): the hot rows' scatter writes those two columns too."""
    rng = np.random.default_rng(5)
    n, s = 500_003, 50_000
    rk = np.unique(rng.integers(-2**63, 2**63 - 1, size=s + 64))[:s]
    rk = rk[rng.permutation(s)]
    lk = rng.integers(-2**63, 2**63 - 1, size=n)
    u = rng.random(n)
    lk[u < 0.4] = rk[rng.integers(0, s, size=int((u < 0.4).sum()))]
    lk[(u >= 0.4) & (u < 0.5)] = rk[7]
    lk[(u >= 0.5) & (u < 0.52)] = rk[8]
    la, ra = rng.integers(-2**31, 2**31, n).astype(np.int32), rng.integers(0, 2**31, s).astype(np.int32)
    t1, t2 = eng.table_from_columns([lk, la]), eng.table_from_columns([rk, ra])
    res = eng.join(t1, t2, 0, 0, [1], [1])
    li, ri = _np_join_rows(lk, rk)
    assert res.shape == (len(li), 2) and eng.last_join_path().startswith("partitioned")
    assert np.array_equal(res.column(0), la[li]) and np.array_equal(res.column(1), ra[ri])
    res.free(); t1.free(); t2.free()


_LONGRUNS = r"""
import sys
import numpy as np
sys.path.insert(0, %r)
sys.path.insert(0, %r)
import test_gpu_hjoin as T
from harkdb_amd.engine import Engine
eng = Engine(0)
rng = np.random.default_rng(31)
for dt, dup in ((np.uint32, False), (np.int64, True), (np.int64, False)):
    # 400 keys with 65 .. 3000 probe rows each: too few for the sample (switched off here), too many for the order kernel's
    # counting: their runs of the stage are sorted in place
    n, s = 900_000, 60_000
    info = np.iinfo(dt)
    rk = rng.integers(info.min, info.max, size=s, dtype=np.int64).astype(dt)
    if dup: rk[:1000] = rk[1000:2000]
    lk = rng.integers(info.min, info.max, size=n, dtype=np.int64).astype(dt)
    at = 0
    for key, rows in zip(rk[rng.choice(s, size=400, replace=False)], rng.integers(65, 1500 if dt is np.int64 else 3000, size=400)):
        lk[rng.integers(0, n, size=int(rows))] = key
    print(dt.__name__, dup, "pairs", T._check(eng, lk, rk), eng.last_join_path())
    assert eng.last_join_path().startswith("partitioned"), eng.last_join_path()
print("long runs ok")
"""


@pytest.mark.parametrize("env", [{"HARK_JOIN_NOHOT": "1"}, {"HARK_JOIN_NOHOT": "1", "HARK_JOIN_STAGE": "4000"}, {"HARK_JOIN_HOTMIN": "3"}])
def test_ranks_with_many_probe_rows_are_sorted_in_the_stage(env):
    """A rank with more than 64 matching probe rows: its run of the order kernel's stage is sorted by a wave (bitonic, in
    LDS) instead of sending ALL survivors through two radix sorts.  With a stage of 4000 survivors some runs are cut by
    sub-round boundaries' neighbours; with a sample threshold of 3 the most frequent of them take the hot path instead."""
    out = subprocess.run([sys.executable, "-c", _LONGRUNS % (ROOT, os.path.join(ROOT, "tests"))], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
    assert out.returncode == 0 and "long runs ok" in out.stdout, out.stdout + out.stderr


@pytest.mark.parametrize("hot", [0, 40, 500, 20_000])
def test_i64_join_of_carried_columns_only(eng, hot):
    """Unique i64 build keys and only the carried probe-side column and the rank-ordered build-side column selected: the
    order kernel writes those two columns and leaves (rank, left row) out.  With a hot key (more probe rows of one key
    than the kernel orders in place: 64) its verdict is "general sort" and it runs once more to deliver the rows."""
    rng = np.random.default_rng(33 + hot)
    n, s = 400_003, 50_000
    rk = np.unique(rng.integers(-2**63, 2**63 - 1, size=s + 64))[:s]
    rk = rk[rng.permutation(s)]
    lk = rng.integers(-2**63, 2**63 - 1, size=n)
    hit = rng.random(n) < 0.5
    lk[hit] = rk[rng.integers(0, s, size=int(hit.sum()))]
    if hot:
        lk[rng.choice(n, size=hot, replace=False)] = rk[123]
    la, ra = rng.integers(-2**31, 2**31, n).astype(np.int32), rng.integers(0, 2**31, s).astype(np.int32)
    t1, t2 = eng.table_from_columns([lk, la]), eng.table_from_columns([rk, ra])
    res = eng.join(t1, t2, 0, 0, [1], [1])
    li, ri = _np_join_rows(lk, rk)
    assert res.shape == (len(li), 2)
    assert np.array_equal(res.column(0), la[li]) and np.array_equal(res.column(1), ra[ri])
    res.free(); t1.free(); t2.free()


_ROUNDS = r"""
import sys
import numpy as np
sys.path.insert(0, %r)
sys.path.insert(0, %r)
import test_gpu_hjoin as T
from harkdb_amd.engine import Engine
eng = Engine(0)
for case in ("u32 every row hits, 8 partners each", "i64 duplicates both sides", "u32 few distinct keys (duplicate splitters)",
             "i64 unique build keys, every second probe row hits", "u32 10%% hits", "i64 consecutive keys among spread ones"):
    lk, rk = T._make(case, 3)
    print(case, T._check(eng, lk, rk))
print("rounds ok")
"""


@pytest.mark.parametrize("chunk", ["7", "64", "120"])
def test_build_slices_longer_than_lds_take_rounds(chunk):
    """HARK_JOIN_CHUNK caps the build keys staged per round (a test knob), so every bucket needs many rounds and runs of
    equal keys cross round boundaries: the results must not change.  i64 buckets of up to two chunks (64: the clustered
    case's ~117 keys, 120: the unique case's ~195) take the single round over truncated keys."""
    env = dict(os.environ, HARK_JOIN_CHUNK=chunk)
    out = subprocess.run([sys.executable, "-c", _ROUNDS % (ROOT, os.path.join(ROOT, "tests"))], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and "rounds ok" in out.stdout, out.stdout + out.stderr


@pytest.mark.parametrize("cap,pieces", [("40", "0"), ("40", "1"), ("700", "0"), ("700", "1")])
def test_rank_order_in_sub_rounds(cap, pieces):
    """HARK_JOIN_STAGE caps the survivors the order kernel stages in LDS per sub-round (a test knob; at full size a
    bucket has far more survivors than the stage holds), so every bucket takes many sub-rounds, and with 40 some groups
    of ranks do not fit at all: the bucket's survivors are regrouped and such a group is taken rank by rank
    (HARK_JOIN_PIECES=1: what a join with many matching rows does; a single rank over the cap goes out unordered and the
    general radix path orders everything) or the whole bucket goes to the general radix path (0): same rows, same order."""
    env = dict(os.environ, HARK_JOIN_STAGE=cap, HARK_JOIN_PIECES=pieces)
    out = subprocess.run([sys.executable, "-c", _ROUNDS % (ROOT, os.path.join(ROOT, "tests"))], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and "rounds ok" in out.stdout, out.stdout + out.stderr


def test_sort_merge_knob_gives_the_same_rows():
    env = dict(os.environ, HARK_JOIN_SORTMERGE="1")
    out = subprocess.run([sys.executable, "-c", _ROUNDS % (ROOT, os.path.join(ROOT, "tests"))], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and "rounds ok" in out.stdout, out.stdout + out.stderr


_NEAR = r"""
import sys
import numpy as np
sys.path.insert(0, %r)
sys.path.insert(0, %r)
import test_gpu_hjoin as T
from harkdb_amd.engine import Engine
eng = Engine(0)
rng = np.random.default_rng(11)
s, n = 60_000, (1 << 18) + 4099
rk = (rng.permutation(s).astype(np.int64) << 36) - (1 << 51)          # unique build keys, 2^36 apart: a bucket's keys truncate to (key - first) >> ts with ts > 0
lk = rk[rng.integers(0, s, size=n)]
near = rng.random(n) < 0.5
lk[near] += rng.integers(1, 1 << 12, size=int(near.sum()))              # half of the probe keys sit just ABOVE a build key: same truncation, no partner
print("pairs", T._check(eng, lk, rk), "of", n, "probe rows;", int((~near).sum()), "expected")
print("near ok")
"""


@pytest.mark.parametrize("chunk", ["64", "100"])
def test_probe_keys_that_share_a_truncation_with_a_build_key_do_not_join(chunk):
    """64-bit keys: a bucket of up to two chunks of build keys is probed in ONE round over keys truncated to 32 bits, and a hit is
    confirmed by the order kernel against the build key of the survivor's rank (k_hjoin.hip).  Probe keys a few units above
    a build key share its truncation: the first attempt's survivors contain rows that do not join, the kernel says so, and
    the join runs again with full keys -- the reference's rows either way (join.fut:52-75)."""
    env = dict(os.environ, HARK_JOIN_CHUNK=chunk)                          # ~117 build keys per bucket: two chunks of 64 / 100 -> the truncated round
    out = subprocess.run([sys.executable, "-c", _NEAR % (ROOT, os.path.join(ROOT, "tests"))], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and "near ok" in out.stdout, out.stdout + out.stderr


def test_partition_kernel_with_compiler_counted_loads_gives_the_same_rows():
    """jpart_kernel issues its loads from inline assembly and waits for them by hand (a software pipeline the compiler's wait
    counting cannot follow); HARK_JOIN_PLAIN_LOADS=1 runs its twin with plain loads.  Every join shape of this file through
    that twin too: a compiler change that breaks the hand-placed waits shows up as a difference between the two."""
    env = dict(os.environ, HARK_JOIN_PLAIN_LOADS="1")
    out = subprocess.run([sys.executable, "-c", _ROUNDS % (ROOT, os.path.join(ROOT, "tests"))], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and "rounds ok" in out.stdout, out.stdout + out.stderr


_CROWD = r"""
import sys
import numpy as np
sys.path.insert(0, %r)
sys.path.insert(0, %r)
import test_gpu_hjoin as T
from harkdb_amd.engine import Engine
eng = Engine(0)
rng = np.random.default_rng(5)
s, n = 512 * 200, 1 << 23          # 16384 probe rows per bucket: far more than a bin's share of the survivor room
for dt in (np.uint32, np.int64):
    rk = np.sort(rng.choice(1 << 30, size=s, replace=False)).astype(dt)          # unique build keys, 200 per bucket
    first = rk.reshape(512, 200)[:, :5].reshape(-1)                              # the five smallest keys of every bucket
    lk = first[rng.integers(0, len(first), size=n)]                              # every probe row hits, all hits crowd a bucket's first ranks
    print(dt.__name__, "pairs", T._check(eng, lk, rk), eng.last_join_path())
    assert eng.last_join_path() == "partitioned", eng.last_join_path()          # the crowded bins spill into the bucket's overflow area
    even = rk[rng.integers(0, s, size=n)]                                        # the same build side probed evenly: the partitioned path
    print(dt.__name__, "pairs", T._check(eng, even, rk), eng.last_join_path())
    assert eng.last_join_path() == "partitioned", eng.last_join_path()
print("crowd ok")
"""


def test_survivor_bin_overflow_goes_to_the_overflow_area():
    """The bucket kernel deals survivors into bins of equal rank ranges with equal room; probe rows that all hit a bucket's
    first few ranks overflow the first bin (HARK_JOIN_STAGE=16 makes the bins small enough at this size): the rest goes to
    the bucket's overflow area, which the order kernel reads in every sub-round -- the reference's rows either way."""
    env = dict(os.environ, HARK_JOIN_STAGE="16", HARK_JOIN_NOHOT="1")     # (the sample would take a part of these keys away)
    out = subprocess.run([sys.executable, "-c", _CROWD % (ROOT, os.path.join(ROOT, "tests"))], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and "crowd ok" in out.stdout, out.stdout + out.stderr


@pytest.mark.parametrize("dt", [np.uint32, np.int64])
def test_probe_rows_crowding_a_stretch_of_the_build_side_cut_the_buckets_by_weight(eng, dt):
    """Two thousand NEIGHBOURING build keys draw 40 % of the probe rows, 100..250 each: no key is frequent enough for the sample
    to call it hot, but the even cut would put them into seventeen of the 512 buckets (their slabs overflow: the sort-merge
    path).  The sampled rows' weight cuts the buckets instead (jhot_select_kernel); with HARK_JOIN_EVEN_CUTS=1 the old cut
    and its fallback -- the reference's rows either way (join.fut:55-75)."""
    rng = np.random.default_rng(17 + (dt is np.int64))
    n, s = 800_000, 60_000
    info = np.iinfo(dt)
    rk = np.unique(rng.integers(info.min, info.max, size=s + 500, dtype=np.int64).astype(dt))[:s]
    crowd = np.sort(rk)[30_000:32_000]
    rk = rk[rng.permutation(s)]
    lk = rng.integers(info.min, info.max, size=n, dtype=np.int64).astype(dt)
    u = rng.random(n)
    lk[u < 0.1] = rk[rng.integers(0, s, size=int((u < 0.1).sum()))]
    m = u > 0.6
    lk[m] = crowd[rng.integers(0, len(crowd), size=int(m.sum()))]
    assert _check(eng, lk, rk) > 300_000
    assert eng.last_join_path() == "partitioned, buckets cut by weight", eng.last_join_path()
    os.environ["HARK_JOIN_EVEN_CUTS"] = "1"
    try:
        assert _check(eng, lk, rk) > 300_000
        assert eng.last_join_path() in ("partitioned", "sort-merge")
    finally:
        del os.environ["HARK_JOIN_EVEN_CUTS"]
    # evenly spread probe keys keep the even cut
    lk2 = rng.integers(info.min, info.max, size=n, dtype=np.int64).astype(dt)
    lk2[u < 0.5] = rk[rng.integers(0, s, size=int((u < 0.5).sum()))]
    assert _check(eng, lk2, rk) > 300_000 and eng.last_join_path() == "partitioned"
