"""GPU parity of projection (select.fut) and WHERE compaction against the oracle."""
import numpy as np
import pytest

from conftest import load_golden, resolve_table

pytestmark = pytest.mark.gpu
OPS = load_golden("operators.json")


@pytest.fixture(scope="module")
def eng():
    from harkdb_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


@pytest.mark.parametrize("case", OPS["query_sel"], ids=lambda c: c["id"])
def test_query_sel_golden(eng, case):
    t = eng.table_from_matrix(resolve_table(case["table"]), np.int32)
    assert eng.query_sel(t, case["cols"]).to_numpy().tolist() == case["out"]


@pytest.mark.parametrize("order", ["C", "F"])
@pytest.mark.parametrize("n", [0, 1, 5, 1023, 100_003])
def test_query_sel_matches_oracle(eng, oracle, n, order):
    rng = np.random.default_rng(n)
    db = np.asarray(rng.integers(-2**31, 2**31, size=(n, 8), dtype=np.int64), order=order)
    t = eng.table_from_matrix(db, np.int32)
    cols = [7, 0, 3, 3]
    assert np.array_equal(eng.query_sel(t, cols).to_numpy(), oracle.query_sel(db, cols))


def test_query_sel_bounds(eng):
    from harkdb_amd._ffi import HarkError, EBOUNDS
    t = eng.table_from_matrix(np.arange(12).reshape(3, 4), np.int32)
    with pytest.raises(HarkError) as ei:
        eng.query_sel(t, [4])
    assert ei.value.code == EBOUNDS


@pytest.mark.parametrize("dtype,value", [(np.float32, 0.5), (np.int32, -3), (np.uint32, 2**31 + 5), (np.int64, 2**40)])
@pytest.mark.parametrize("cmp", [">", ">=", "<", "<=", "=", "!="])
@pytest.mark.parametrize("n", [0, 1, 4097, 250_001])
def test_filter_indices_bit_exact(eng, oracle, dtype, value, cmp, n):
    rng = np.random.default_rng(n + 17)
    if dtype == np.float32:
        col = rng.random(n, dtype=np.float32)
        col[::5] = 0.5
    elif dtype == np.int32:
        col = rng.integers(-10, 10, size=n).astype(np.int32)
    elif dtype == np.uint32:
        col = (rng.integers(0, 12, size=n) + 2**31).astype(np.uint32)
    else:
        col = (rng.integers(-4, 4, size=n) + 2**40).astype(np.int64)
    other = rng.integers(-2**31, 2**31, size=n).astype(np.int32)
    other64 = rng.integers(-2**62, 2**62, size=n).astype(np.int64)
    t = eng.table_from_columns([col, other, other64])
    res = eng.filter_sel(t, 0, cmp, value, [1, 2, 0])
    idx = oracle.filter_indices(col, cmp, value)
    assert np.array_equal(res.column(0), idx)             # compaction indices: bit-exact
    assert np.array_equal(res.column(1), other[idx])
    assert np.array_equal(res.column(2), other64[idx])
    assert np.array_equal(res.column(3), col[idx])


def test_filter_all_and_none(eng):
    col = np.arange(10_000, dtype=np.float32)
    t = eng.table_from_columns([col])
    assert eng.filter_sel(t, 0, ">=", 0.0, [0]).shape == (10_000, 2)
    assert eng.filter_sel(t, 0, "<", 0.0, [0]).shape == (0, 2)


@pytest.mark.parametrize("n", [0, 4, 1000, 1 << 20, (1 << 22) + 12])
@pytest.mark.parametrize("nbuf", [1, 2, 3])
def test_stream_read_probe_reads_every_byte(eng, n, nbuf):
    """bench.py's bandwidth probe: the 64-bit XOR fold proves every 16-byte word of every buffer was read once."""
    rng = np.random.default_rng(n + nbuf)
    bufs, exp = [], 0
    fold = eng.alloc(8)
    eng.upload(fold, np.zeros(1, dtype=np.uint64))
    for _ in range(nbuf):
        a = rng.integers(0, 2**32, size=n, dtype=np.uint32)
        b = eng.alloc(max(n * 4, 16))
        if n:
            eng.upload(b, a)
            q = a.reshape(-1, 4)
            exp ^= (int(np.bitwise_xor.reduce(q[:, 0] ^ q[:, 2])) << 32) | int(np.bitwise_xor.reduce(q[:, 1] ^ q[:, 3]))
        bufs.append(b)
    eng.stream_read(bufs, n * 4, fold)
    assert int(eng.download(fold, 1, np.uint64)[0]) == exp
    for b in bufs + [fold]:
        eng.free(b)
