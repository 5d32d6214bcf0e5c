"""TWO ranks on ONE GPU (gloo carries the collectives, staged through the host; RCCL refuses two ranks on one
device): the sharded SQL surface and the fused sharded operator with real kernels on every rank -- range / hash
partition on the device, all-to-all, second-level aggregation, sample sort, all-reduce of the accumulators.
The RCCL transport itself is covered with one rank in test_gpu_sharded.py; N > 1 GPUs are the driver's."""
import os
import socket
import sys

import numpy as np
import pandas as pd
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STATEMENTS = [
    "select k, w from t where p > 0.9 limit 100",
    "select k, sum(v), count(*), avg(v), min(w) from t where p > 0.5 group by k having count(*) > 500 order by sum(v) desc limit 7",
    "select s, sum(v), count(*), max(k) from t where p > 0.25 group by s",
    "select k, w from t where p > 0.5 order by k",
    "select w, k from t order by p desc limit 50",
    "select s, v from t where p > 0.99 order by s desc",
    "select t.k, b.y, t.w from t join b on t.w = b.x",
    "select w, k, sum(v), count(*) from t where p > 0.5 group by k, w",
    "select k, s, max(w), avg(v) from t group by k, s having count(*) > 1 order by max(w) desc limit 40",
    # dense key domains: per-slot partials merged by all-reduce, the tail on the device
    "select d, sum(v), count(*), avg(v), min(w), max(p) from t where p > 0.5 group by d having count(*) > 3 order by sum(v) desc limit 10",
    "select d, sum(w), min(v), count(*) from t group by d",
    "select d, e, sum(v), count(*) from t where p > 0.25 group by d, e order by count(*) desc limit 25",
    "select d, max(w) from t group by d having max(w) > 7 order by d desc",
    # ORDER BY on several keys over shards (composite key with the key ranges of all shards)
    "select k, w, v from t where p > 0.7 order by k, w",
    "select w, k from t order by w desc, k desc limit 60",
    # the join in the reference's order, with a LIMIT cut
    "select t.k, b.y, t.w from t join b on t.w = b.x limit 1000",
    # clauses around a join: conjuncts below the exchange, the owners' pairs as a sharded table for the rest
    "select b.y, sum(v), count(*) from t join b on t.w = b.x where p > 0.5 and y < 600 group by b.y having count(*) > 10 order by sum(v) desc limit 12",
    "select t.k, b.y from t join b on t.w = b.x where p > 0.98 order by y desc limit 40",
]
DENSE = {9, 10, 11, 12}          # statements whose aggregation must take the all-reduce path


def _frames():
    rng = np.random.default_rng(1)
    n = 60_000
    df = pd.DataFrame({"k": rng.integers(-20, 20, n).astype(np.int32), "p": rng.random(n).astype(np.float32),
                       "v": rng.integers(0, 16, n).astype(np.float32), "w": rng.integers(-9, 9, n).astype(np.int32),
                       "s": (rng.integers(0, 3000, n) * 1_000_003 % (2**31)).astype(np.int32),
                       "d": rng.integers(0, 5000, n).astype(np.int32), "e": rng.integers(3, 9, n).astype(np.int32)})
    df.loc[: n // 2, "d"] %= 2500                       # keys 2500.. exist on the second shard only
    b = pd.DataFrame({"x": rng.integers(-12, 12, 700).astype(np.int32), "y": rng.integers(0, 1000, 700).astype(np.int32)})
    return df, b


def _worker(rank, world, port, q, csv_path, txt_path):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      HARK_DIST_BACKEND="gloo")
    import torch
    import torch.distributed as dist
    from harkdb_amd import dist as hd
    from harkdb_amd.engine import FgbPlan
    c = hd.ShardedFutharkContext()
    df, b = _frames()
    # three ways in: a CSV file cut into byte ranges (nobody parses the whole file), this rank's rows only, device columns
    c.create_table("t", csv_path)
    lo, hi = c.rows["t"][1], c.rows["t"][2]
    assert c.rows["t"][0] == len(df) and c.local.tables["t"]._device.shape[0] == hi - lo < len(df)
    c.create_table("t_local", df.iloc[lo:hi], local=True)
    assert c.rows["t_local"] == c.rows["t"]
    c.create_table("b", b)
    c.create_table("tx", txt_path)                      # a TXT file: sliced per rank under the whole table's headers c1, c2, ...
    assert c.local.tables["tx"].get_schema() == ["c1", "c2", "c3"] and c.rows["tx"][0] == TXT_ROWS
    xs = torch.arange(1000 * rank, 1000 * rank + 1000 + 24 * rank, dtype=torch.int32, device=c.device)
    ys = (xs % 7).to(torch.float32)
    c.create_table_from_device("g", ["x", "y"], [xs.data_ptr(), ys.data_ptr()], [np.int32, np.float32], xs.numel(), keepalive=(xs, ys))
    assert c.rows["g"][0] == sum(1000 + 24 * r for r in range(world)) and c.rows["g"][1] == sum(1000 + 24 * r for r in range(rank))
    out = {}
    paths = {}
    for j, stmt in enumerate(STATEMENTS):
        names, cols = c.sql_columns(stmt)
        out[stmt] = (names, [np.asarray(x) for x in cols])
        paths[j] = getattr(c, "last_groupby_path", None) if " group by " in stmt else None
    out["late"] = getattr(c, "late_aggregations", 0)
    out["paths"] = paths
    out["g"] = c.sql_columns("select x, y from g where y > 5 order by x desc limit 30")
    out["t_local"] = c.sql_columns("select d, count(*) from t_local group by d")
    out["tx"] = c.sql_columns("select c1, c3 from tx where c2 > 10")
    # the fused operator over shards: local kernels, all-reduce of the accumulators, finish
    G, n = 1 << 14, len(df)
    lo, hi = hd.shard_range(n, rank, world)
    eng = c.local.FutEnv
    kk = (np.arange(n, dtype=np.int64) * 2654435761 % G).astype(np.int32)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(c.device)
    p, k, v = dev(df.p.to_numpy()[lo:hi]), dev(kk[lo:hi]), dev(df.v.to_numpy()[lo:hi])
    plan = FgbPlan(eng, hi - lo, G)
    job = hd.ShardedFgb(eng, plan, c.device)
    so, co = torch.empty(G, dtype=torch.float32, device=c.device), torch.empty(G, dtype=torch.int64, device=c.device)
    job.step(p.data_ptr(), ">", 0.5, k.data_ptr(), v.data_ptr(), hi - lo, so.data_ptr(), co.data_ptr())
    torch.cuda.synchronize()
    out["fgb"] = (so.cpu().numpy(), co.cpu().numpy())
    q.put((rank, out))
    dist.barrier(); dist.destroy_process_group()


TXT_ROWS = 5001


def _txt_matrix():
    return np.random.default_rng(5).integers(-30, 30, size=(TXT_ROWS, 3))


@pytest.mark.parametrize("world", [2, 3, 4])
def test_ranks_on_one_gpu_match_single_context(tmp_path, world):
    """Two, three and four ranks (uneven shards, up to three splitters in the range partitions) sharing one GPU."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    df, b = _frames()
    csv_path = str(tmp_path / "t.csv")
    df.to_csv(csv_path, index=False)
    txt_path = str(tmp_path / "tx.txt")
    np.savetxt(txt_path, _txt_matrix(), fmt="%d")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, csv_path, txt_path)) for r in range(world)]
    for pr in procs:
        pr.start()
    outs = dict(q.get(timeout=300) for _ in range(world))
    for pr in procs:
        pr.join(timeout=120)
        assert pr.exitcode == 0
    # single-context reference on the same GPU
    from harkdb_amd.context import FutharkContext
    fc = FutharkContext(device=0, sql_mode=True)
    fc.create_table("t", df)
    fc.create_table("b", b)
    fc.create_table("tx", txt_path)
    nx, cx = fc.sql_columns("select c1, c3 from tx where c2 > 10")
    for rank in range(world):
        gn, gc = outs[rank]["tx"]
        assert gn == nx and all(np.array_equal(np.asarray(x), np.asarray(y)) for x, y in zip(gc, cx))
    for stmt in STATEMENTS:
        names, cols = fc.sql_columns(stmt)
        for rank in range(world):
            gn, gc = outs[rank][stmt]
            assert gn == names, stmt
            for x, y in zip(gc, cols):                            # row for row, the join included (reference order, join.fut:52-75)
                assert x.dtype == y.dtype and len(x) == len(y), stmt
                if x.dtype.kind == "f":
                    assert np.allclose(x, y, rtol=1e-6), stmt
                else:
                    assert np.array_equal(x, y), stmt
    for rank in range(world):
        for j in DENSE:
            assert outs[rank]["paths"][j] == "dense all-reduce", (j, outs[rank]["paths"])
        assert outs[rank]["paths"][1] == "owner all-to-all"      # negative keys: partial aggregates travel to the owner of hash(key)
        assert outs[rank]["late"] >= 2                           # statements 1 and 9: MIN / MAX nobody orders by, for the LIMIT groups only, merged over the shards
        gx = np.concatenate([np.arange(1000 * r, 1000 * r + 1000 + 24 * r) for r in range(world)]).astype(np.int32)
        gx = np.sort(gx[(gx % 7) > 5], kind="stable")[::-1][:30]
        assert np.array_equal(outs[rank]["g"][1][0], gx) and np.array_equal(outs[rank]["g"][1][1], (gx % 7).astype(np.float32))
        ld, lc = outs[rank]["t_local"][1]
        ed, ec = np.unique(df.d.to_numpy(), return_counts=True)
        assert np.array_equal(ld, ed.astype(np.int32)) and np.array_equal(lc, ec)
    # the join against the oracle's reference order as well (ascending u32 key, left row, right row; 18 distinct keys over
    # 60 000 rows: every key's rows sit on both shards and meet at ONE owner)
    from oracle import oracle as ora
    ref = ora.join(df[["k", "w"]].to_numpy().astype(np.int64), b[["x", "y"]].to_numpy().astype(np.int64), 1, 0, [0, 1], [1])
    for rank in range(world):
        got = outs[rank]["select t.k, b.y, t.w from t join b on t.w = b.x"][1]
        mine = np.ascontiguousarray(np.stack([got[0], got[2], got[1]], axis=1).astype(np.int32)).view(np.uint32)
        assert np.array_equal(mine, ref)
    n, G = len(df), 1 << 14
    kk = (np.arange(n, dtype=np.int64) * 2654435761 % G).astype(np.int32)
    keep = df.p.to_numpy() > 0.5
    es = np.bincount(kk[keep], weights=df.v.to_numpy()[keep].astype(np.float64), minlength=G).astype(np.float32)
    ec = np.bincount(kk[keep], minlength=G).astype(np.int64)
    for rank in range(world):
        assert np.array_equal(outs[rank]["fgb"][0], es) and np.array_equal(outs[rank]["fgb"][1], ec)
