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
        assert np.array_equal(out[np.lexsort(out.T[::-1])], ref[np.lexsort(ref.T[::-1])].astype(out.dtype))   # same multiset of rows
    finally:
        sfc.device_exchange = False
