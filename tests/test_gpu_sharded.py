"""ShardedFutharkContext on one GPU (world size 1 under RCCL): the sharded
code path (shard, local operators, merge) must reproduce the single-context
results.  World size 2 logic is covered on CPU by tests/test_dist_gloo.py."""
import os
import socket

import numpy as np
import pandas as pd
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sfc():
    import torch.distributed as dist
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from harkdb_amd.dist import ShardedFutharkContext
    c = ShardedFutharkContext()
    rng = np.random.default_rng(1)
    n = 50_000
    df = pd.DataFrame({"k": rng.integers(-20, 20, n).astype(np.int32), "p": rng.random(n).astype(np.float32),
                       "v": rng.integers(0, 16, n).astype(np.float32), "w": rng.integers(-9, 9, n).astype(np.int32)})
    c.create_table("t", df)
    c._df = df
    yield c
    if dist.is_initialized():
        dist.destroy_process_group()
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)


def test_sharded_where_projection(sfc):
    df = sfc._df
    out = sfc.sql("select k, w from t where p > 0.9 limit 100")
    assert np.array_equal(out, df[df.p > 0.9][["k", "w"]].head(100).to_numpy())


def test_sharded_groupby_having_order(sfc):
    df = sfc._df
    names, cols = sfc.sql_columns("select k, sum(v), count(*), avg(v), min(w) from t where p > 0.5 group by k "
                                  "having count(*) > 500 order by sum(v) desc limit 7")
    g = df[df.p > 0.5].groupby("k").agg(s=("v", "sum"), c=("v", "count"), a=("v", "mean"), m=("w", "min")).reset_index()
    g = g[g.c > 500].sort_values("s", ascending=False, kind="stable").head(7)
    assert names == ["k", "sum(v)", "count(*)", "avg(v)", "min(w)"]
    assert np.array_equal(cols[0], g.k.to_numpy()) and np.array_equal(cols[1], g.s.to_numpy().astype(np.float32))
    assert np.array_equal(cols[2], g.c.to_numpy()) and np.allclose(cols[3], g.a.to_numpy(), rtol=1e-6)
    assert np.array_equal(cols[4], g.m.to_numpy())


def test_sharded_device_exchange_paths(sfc, oracle):
    """The RCCL all-to-all repartition (hash partition + gather on the GPU, all_to_all_single,
    second-level aggregation / local join), forced on with one rank."""
    df = sfc._df
    sfc.device_exchange = True
    try:
        names, cols = sfc.sql_columns("select w, sum(v), count(*), max(k), avg(v) from t where p > 0.25 group by w")
        g = df[df.p > 0.25].groupby("w").agg(s=("v", "sum"), c=("v", "count"), m=("k", "max"), a=("v", "mean")).reset_index()
        assert np.array_equal(cols[0], g.w.to_numpy()) and np.array_equal(cols[1], g.s.to_numpy().astype(np.float32))
        assert np.array_equal(cols[2], g.c.to_numpy()) and np.array_equal(cols[3], g.m.to_numpy())
        assert np.allclose(cols[4], g.a.to_numpy(), rtol=1e-6)
        rng = np.random.default_rng(3)
        a = rng.integers(0, 40, size=(3000, 3)).astype(np.int64)
        b = rng.integers(0, 40, size=(500, 2)).astype(np.int64)
        sfc.create_table("a", a); sfc.create_table("b", b)
        out = sfc.sql("select a.col1, b.col2, a.col3 from a join b on a.col2 = b.col1")
        ref = oracle.join(a, b, 1, 0, [0, 2], [1])[:, [0, 2, 1]]
        assert np.array_equal(out, ref.astype(out.dtype))           # row for row: the reference's order (key, left row, right row)
    finally:
        sfc.device_exchange = False


@pytest.mark.parametrize("dtype", [np.uint32, np.int32, np.float32, np.int64])
@pytest.mark.parametrize("descending", [False, True])
def test_partition_by_range_matches_numpy(sfc, dtype, descending):
    """hark_op_partition_by_range: part = number of splitters <= key in the column's own order."""
    import torch
    eng = sfc.local.FutEnv
    rng = np.random.default_rng(5)
    n = 70_001
    if dtype == np.float32:
        keys = (rng.standard_normal(n) * 10).astype(np.float32)
        keys[:7] = [0.0, -0.0, np.inf, -np.inf, 1.5, -1.5, 0.0]
    elif dtype == np.int64:
        keys = rng.integers(-2**40, 2**40, n).astype(np.int64)
    elif dtype == np.uint32:
        keys = rng.integers(0, 2**32, n).astype(np.uint32)
    else:
        keys = rng.integers(-2**31, 2**31, n).astype(np.int32)
    splitters = np.sort(keys[rng.integers(0, n, 5)])
    tdt = {np.uint32: torch.int32, np.int32: torch.int32, np.float32: torch.float32, np.int64: torch.int64}[dtype]
    kdev = torch.from_numpy(keys.view(np.int32) if dtype == np.uint32 else keys).to(sfc.device).view(tdt)
    perm = torch.empty(n, dtype=torch.int32, device=sfc.device)
    counts = eng.partition_by_range(kdev.data_ptr(), dtype, n, splitters, descending, perm.data_ptr())
    d = np.searchsorted(splitters, keys, side="right")
    dest = (len(splitters) - d) if descending else d
    assert counts == np.bincount(dest, minlength=len(splitters) + 1).tolist()
    assert np.array_equal(perm.cpu().numpy().view(np.uint32), np.argsort(dest, kind="stable").astype(np.uint32))
    # no splitters: one part, identity permutation
    counts1 = eng.partition_by_range(kdev.data_ptr(), dtype, n, splitters[:0], descending, perm.data_ptr())
    assert counts1 == [n] and np.array_equal(perm.cpu().numpy(), np.arange(n, dtype=np.int32))


def test_sharded_orderby_sample_sort(sfc):
    """ORDER BY over shards (sample sort with the RCCL all-to-all, forced on with one rank)
    equals the single-context result, ties in table order."""
    df = sfc._df
    sfc.device_exchange = True
    try:
        for stmt, ref in (
            ("select k, w from t where p > 0.5 order by k", df[df.p > 0.5].sort_values("k", kind="stable")[["k", "w"]]),
            ("select w, k from t order by p desc limit 50", df.sort_values("p", ascending=False, kind="stable")[["w", "k"]].head(50)),
            ("select k from t where p > 2.0 order by k", df[df.p > 2.0][["k"]]),
        ):
            names, cols = sfc.sql_columns(stmt)
            assert names == list(ref.columns)
            for c, name in zip(cols, names):
                assert np.array_equal(c, ref[name].to_numpy())
            loc = sfc.local.sql_columns(stmt)[1]
            assert all(np.array_equal(a, b) for a, b in zip(cols, loc))
    finally:
        sfc.device_exchange = False


def test_sharded_multi_key_groupby(sfc):
    """GROUP BY on several keys over shards: the composite key is encoded with the all-rank ranges; both merge paths."""
    df = sfc._df
    g = df[df.p > 0.5].groupby(["w", "k"]).agg(s=("v", "sum"), n=("v", "count")).reset_index()
    for exchange in (False, True):
        sfc.device_exchange = exchange
        try:
            names, cols = sfc.sql_columns("select k, w, sum(v), count(*) from t where p > 0.5 group by w, k")
        finally:
            sfc.device_exchange = False
        assert names == ["k", "w", "sum(v)", "count(*)"]
        assert np.array_equal(cols[0], g.k.to_numpy()) and np.array_equal(cols[1], g.w.to_numpy())
        assert np.array_equal(cols[2], g.s.to_numpy().astype(np.float32)) and np.array_equal(cols[3], g.n.to_numpy())
    assert "__mk_t" not in sfc.local.tables
