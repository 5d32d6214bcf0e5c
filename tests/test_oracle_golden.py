"""The CPU oracle against every golden vector the reference holds for the path
(the 18 KATs of segmented_tests.fut) and the hand-derived operator goldens
G1-G9."""
import numpy as np
import pytest

from conftest import load_golden, resolve_table, expand_runs

KAT = load_golden("segmented_kat.json")
OPS = load_golden("operators.json")


@pytest.mark.parametrize("case", KAT["segmented_scan"], ids=lambda c: c["ref"])
def test_segmented_scan(oracle, case):
    assert oracle.segmented_scan_add(case["flags"], case["as"]).tolist() == case["out"]


@pytest.mark.parametrize("case", KAT["segmented_reduce"], ids=lambda c: c["ref"])
def test_segmented_reduce(oracle, case):
    assert oracle.segmented_reduce_add(case["flags"], case["as"]).tolist() == case["out"]


@pytest.mark.parametrize("case", KAT["replicated_iota"], ids=lambda c: c["ref"])
def test_replicated_iota(oracle, case):
    assert oracle.replicated_iota(case["in"]).tolist() == case["out"]


@pytest.mark.parametrize("case", KAT["segmented_iota"], ids=lambda c: c["ref"])
def test_segmented_iota(oracle, case):
    assert oracle.segmented_iota(case["flags"]).tolist() == case["out"]


@pytest.mark.parametrize("case", KAT["expand"], ids=lambda c: c["ref"])
def test_expand(oracle, case):
    assert oracle.test_expand(case["in"]).tolist() == case["out"]


@pytest.mark.parametrize("case", KAT["expand_reduce"], ids=lambda c: c["ref"])
def test_expand_reduce(oracle, case):
    assert oracle.test_expand_reduce(case["in"]).tolist() == case["out"]


@pytest.mark.parametrize("case", KAT["expand_outer_reduce"], ids=lambda c: c["ref"])
def test_expand_outer_reduce(oracle, case):
    assert oracle.test_expand_outer_reduce(case["in"]).tolist() == case["out"]


def test_kat_count():
    assert sum(len(v) for k, v in KAT.items() if not k.startswith("_")) == 18


@pytest.mark.parametrize("case", OPS["query_sel"], ids=lambda c: c["id"])
def test_query_sel(oracle, case):
    out = oracle.query_sel(resolve_table(case["table"]), case["cols"])
    assert out.tolist() == case["out"]


@pytest.mark.parametrize("case", OPS["query_groupby"], ids=lambda c: c["id"])
def test_query_groupby(oracle, case):
    out = oracle.query_groupby(resolve_table(case["table"]), case["g_col"], case["s_cols"], case["t_cols"])
    assert out.tolist() == case["out"]


@pytest.mark.parametrize("case", OPS["join"], ids=lambda c: c["id"])
def test_join(oracle, case):
    out = oracle.join(resolve_table(case["db1"]), resolve_table(case["db2"]), case["col1"], case["col2"],
                      case["cols1"], case["cols2"])
    w = len(case["cols1"]) + len(case["cols2"])
    exp = expand_runs(case["out_runs"], w) if "out_runs" in case else np.asarray(case["out"], dtype=np.int64).reshape(-1, w)
    assert out.astype(np.int64).tolist() == exp.tolist()


def test_groupby_matches_numpy_model(oracle):
    """Property check of the restatement against an independent numpy model
    (sort-free): per distinct key, fold rows in table order."""
    rng = np.random.default_rng(7)
    db = rng.integers(0, 2**32, size=(500, 5), dtype=np.uint64).astype(np.uint32)
    db[:, 1] = rng.integers(0, 13, size=500)
    db[::7, 1] = 0xFFFFFFF0 + (db[::7, 1] & 3)      # keys with the sign bit set
    out = oracle.query_groupby(db, 1, [0, 2, 3, 4, 1], [2, 1, 3, 4, 0])
    keys = np.unique(db[:, 1])
    assert out[:, 0].tolist() == keys.tolist()      # ascending unsigned
    for r, key in zip(out, keys):
        rows = db[db[:, 1] == key].astype(np.uint64)
        assert r[1] == int(rows[:, 0].sum()) & 0xFFFFFFFF
        prod = 1
        for x in rows[:, 2]:
            prod = (prod * int(x)) & 0xFFFFFFFF
        assert r[2] == prod
        assert r[3] == rows[:, 3].max()
        assert r[4] == rows[:, 4].min()
        assert r[5] == key                           # opcode 0 -> min of the key column


def test_bounds_errors(oracle):
    db = np.arange(12).reshape(3, 4)
    db[1, 0] = 0                                     # duplicate key so `merge` actually runs
    with pytest.raises(oracle.OracleError):
        oracle.query_sel(db, [4])
    with pytest.raises(oracle.OracleError):
        oracle.query_groupby(db, 0, [1, 2], [2])     # t_cols shorter than s_cols
    with pytest.raises(oracle.OracleError):
        oracle.query_groupby(db, 9, [1], [2])


def test_join_matches_numpy_model(oracle):
    rng = np.random.default_rng(11)
    a = rng.integers(0, 9, size=(40, 3)).astype(np.uint32)
    b = rng.integers(0, 9, size=(25, 2)).astype(np.uint32)
    out = oracle.join(a, b, 1, 0, [0, 2], [1, 0])
    exp = []
    for key in np.unique(np.concatenate([a[:, 1], b[:, 0]])):
        for i in np.nonzero(a[:, 1] == key)[0]:
            for j in np.nonzero(b[:, 0] == key)[0]:
                exp.append([a[i, 0], a[i, 2], b[j, 1], b[j, 0]])
    assert out.tolist() == np.asarray(exp, dtype=np.uint32).reshape(-1, 4).tolist()


def test_cpu_baselines_agree_with_dense_oracle(oracle):
    """bench.py's two CPU baselines (the reference-algorithm port and the
    multi-threaded single-pass aggregate) compute the same query as the dense
    oracle the GPU is checked against (exact-valued workload: sums are exact)."""
    G = 1 << 12
    p, k, v = oracle.gen_columns(7, 0, 200_000, G, True)
    s32, s64, cnt = oracle.filter_groupby_dense_f32(p, k, v, ">", 0.5, G)
    keys, sums, counts = oracle.filter_groupby_refalgo_f32(p, k, v, ">", 0.5)
    live = np.nonzero(cnt)[0]
    assert np.array_equal(keys, live.astype(np.uint32))
    assert np.array_equal(counts.astype(np.int64), cnt[live])
    assert np.array_equal(sums.astype(np.float64), s64[live])
    for threads in (1, 3, 8):
        m64, mcnt = oracle.filter_groupby_dense_f32_mt(p, k, v, ">", 0.5, G, threads)
        assert np.array_equal(mcnt, cnt) and np.array_equal(m64, s64)
    with pytest.raises(oracle.OracleError):
        oracle.filter_groupby_dense_f32_mt(p, k, v, ">", 0.5, G // 2, 2)
