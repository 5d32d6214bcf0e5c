"""ORDER BY on i64 keys, large tables: the most-significant-digit-first tuple sort (harkdb_amd/csrc/k_msort.hip) against numpy's
stable argsort -- whatever the distribution, the result is the one the tuple passes give (the reference's rsort is stable:
futhark/groupby.fut:8-22), and what does not fit the buckets falls back to them."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from harkdb_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def make_keys(kind, n, rng):
    if kind == "spread":
        return rng.integers(-2**63, 2**63 - 1, size=n)
    if kind == "sorted":                    # every workgroup's rows go to ONE level-1 bucket
        return np.sort(rng.integers(-2**62, 2**62, size=n))
    if kind == "reversed":
        return np.sort(rng.integers(-2**62, 2**62, size=n))[::-1].copy()
    if kind == "positive_48":               # a range of 2^48: the map shifts by 16 bits
        return rng.integers(0, 2**48, size=n)
    if kind == "range_2_33":                # barely over 2^32: one reduced key per two keys
        return rng.integers(-2**32, 2**32, size=n)
    if kind == "some_equal":                # equal keys: ties go by row id
        k = rng.integers(-2**62, 2**62, size=n)
        k[rng.integers(0, n, size=n // 8)] = k[rng.integers(0, n, size=n // 8)]
        return k
    if kind == "few_distinct":              # heavy duplicates: a final bucket overflows -> the tuple passes
        pool = rng.integers(-2**62, 2**62, size=300)
        return pool[rng.integers(0, 300, size=n)]
    if kind == "normal":                    # a lumpy distribution: the affine map overfills the middle buckets, the equalised map does not
        return (rng.standard_normal(n) * 2.0**55).astype(np.int64)
    if kind == "exponential":               # a long tail on one side: the equalised map
        return (rng.exponential(size=n) * 2.0**52).astype(np.int64)
    if kind == "outliers":                  # two keys far outside everyone else's range (the sample may or may not see them)
        k = rng.integers(-2**40, 2**40, size=n)
        k[n // 3] = 2**62; k[n // 2] = -2**62
        return k
    if kind == "clusters":                  # tight clusters around a few centres
        c = rng.integers(-2**62, 2**62, size=64)
        return c[rng.integers(0, 64, size=n)] + rng.integers(0, 2**20, size=n)
    raise ValueError(kind)


KINDS = ["spread", "sorted", "reversed", "positive_48", "range_2_33", "some_equal", "few_distinct", "normal", "exponential", "outliers", "clusters"]


@pytest.mark.parametrize("n", [(1 << 20) + 777, 3_000_001])
@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("descending", [False, True])
def test_large_i64_sort_matches_numpy(eng, n, kind, descending):
    rng = np.random.default_rng(len(kind) * 31 + n % 97 + int(descending))
    key = make_keys(kind, n, rng).astype(np.int64)
    val = rng.integers(-2**31, 2**31, size=n).astype(np.int32)
    t = eng.table_from_columns([key, val])
    res = eng.sort(t, 0, [0, 1], descending=descending)
    perm = np.argsort(~key if descending else key, kind="stable")
    assert np.array_equal(res.column(0), key[perm])
    assert np.array_equal(res.column(1), val[perm])
    res.free(); t.free()


def test_large_i64_sort_same_with_and_without_the_msd_path(eng, monkeypatch):
    rng = np.random.default_rng(5)
    n = 2_500_003
    key = rng.integers(-2**63, 2**63 - 1, size=n).astype(np.int64)
    key[::7] = key[3]
    rowid = np.arange(n, dtype=np.int32)
    t = eng.table_from_columns([key, rowid])
    a = eng.sort(t, 0, [1, 0])
    monkeypatch.setenv("HARK_SORT_NO_MSD", "1")
    b = eng.sort(t, 0, [1, 0])
    assert np.array_equal(a.column(0), b.column(0)) and np.array_equal(a.column(1), b.column(1))
    assert np.array_equal(a.column(0), rowid[np.argsort(key, kind="stable")])
    a.free(); b.free(); t.free()


def test_the_paths_the_sort_reports(tmp_path):
    """Which map took a column, and whether the three sweeps finished, is said on stderr under HARK_SORT_MSD_VERBOSE: uniform keys take
    the affine map, normally distributed keys the equalised one, a column of 300 distinct keys gives up (and the tuple passes sort
    it, and the column remembers: hark_column::msd_unfit) -- results are checked above, this keeps the fast paths from turning into
    fall-backs unnoticed."""
    import os, subprocess, sys, textwrap
    from conftest import ROOT
    script = tmp_path / "paths.py"
    script.write_text(textwrap.dedent("""
        import sys, numpy as np
        sys.path.insert(0, %r)
        from harkdb_amd.engine import Engine
        eng = Engine(0)
        rng = np.random.default_rng(1)
        n = 3_000_001
        pool = rng.integers(-2**62, 2**62, size=300)
        for name, key in (("uniform", rng.integers(-2**62, 2**62, size=n)), ("normal", (rng.standard_normal(n) * 2.0**55).astype(np.int64)),
                          ("few", pool[rng.integers(0, 300, size=n)])):
            print("column", name, file=sys.stderr, flush=True)
            t = eng.table_from_columns([key.astype(np.int64), np.arange(n, dtype=np.int32)])
            eng.sort(t, 0, [0, 1]).free()
            print("again", name, file=sys.stderr, flush=True)
            eng.sort(t, 0, [0, 1]).free(); t.free()
        """ % ROOT))
    env = dict(os.environ, HARK_SORT_MSD_VERBOSE="1")
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    said, again = {}, {}
    cur = None
    for line in r.stderr.splitlines():
        if line.startswith("column "):
            cur, second = line.split()[1], False
        elif line.startswith("again "):
            second = True
        elif line.startswith("msd sort:") and cur:
            (again if second else said)[cur] = line
    assert "gave_up=0" in said["uniform"] and "equalised=0" in said["uniform"], said
    assert "gave_up=0" in said["normal"] and "equalised=1" in said["normal"], said
    assert "gave_up=0" not in said["few"], said
    # the verdict stays with the column: the second ORDER BY on it does not try the three sweeps again; the others do
    assert "few" not in again and "uniform" in again and "normal" in again, again


@pytest.mark.parametrize("n", [110_000_003, 200_000_001])
def test_tables_beyond_1e8_rows_take_the_wide_final_workgroups(eng, n):
    """Up to ~1.05e8 rows the final buckets (<= 2560 tuples) are sorted by workgroups of 512 threads; up to ~2.1e8 rows by workgroups
    of 1024 threads (<= 5120 tuples).  The permutation must be torch's stable argsort exactly."""
    import torch
    from harkdb_amd.dist import tensor_from_ptr
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(n % 1000)
    keys = torch.randint(-2**62, 2**62, (n,), dtype=torch.int64, device=dev, generator=g)
    dup = torch.randint(0, n, (n // 1000,), dtype=torch.int64, device=dev, generator=g)
    keys[dup] = keys[(dup * 7919) % n]                                     # some equal keys, here and there: ties go by row id
    rid = torch.arange(n, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    t = eng.table_from_device(n, [keys.data_ptr(), rid.data_ptr()], [np.int64, np.int32], keepalive=(keys, rid))
    res = eng.sort(t, 0, [0, 1])
    got_k = tensor_from_ptr(res.device_ptr(0), n, np.int64, dev)
    got_r = tensor_from_ptr(res.device_ptr(1), n, np.int32, dev)
    want = torch.sort(keys, stable=True)
    assert torch.equal(got_k, want.values)
    assert torch.equal(got_r.to(torch.int64), want.indices)
    del want
    res.free(); t.free()
