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
]


def _frames():
    rng = np.random.default_rng(1)
    n = 60_000
    df = pd.DataFrame({"k": rng.integers(-20, 20, n).astype(np.int32), "p": rng.random(n).astype(np.float32),
                       "v": rng.integers(0, 16, n).astype(np.float32), "w": rng.integers(-9, 9, n).astype(np.int32),
                       "s": (rng.integers(0, 3000, n) * 1_000_003 % (2**31)).astype(np.int32)})
    b = pd.DataFrame({"x": rng.integers(-12, 12, 700).astype(np.int32), "y": rng.integers(0, 1000, 700).astype(np.int32)})
    return df, b


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      HARK_DIST_BACKEND="gloo")
    import torch
    import torch.distributed as dist
    from harkdb_amd import dist as hd
    from harkdb_amd.engine import FgbPlan
    c = hd.ShardedFutharkContext()
    df, b = _frames()
    c.create_table("t", df)
    c.create_table("b", b)
    out = {}
    for stmt in STATEMENTS:
        names, cols = c.sql_columns(stmt)
        out[stmt] = (names, [np.asarray(x) for x in cols])
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


def test_two_ranks_one_gpu_match_single_context():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    outs = dict(q.get(timeout=300) for _ in range(2))
    for pr in procs:
        pr.join(timeout=120)
        assert pr.exitcode == 0
    # single-context reference on the same GPU
    from harkdb_amd.context import FutharkContext
    df, b = _frames()
    fc = FutharkContext(device=0, sql_mode=True)
    fc.create_table("t", df)
    fc.create_table("b", b)
    for stmt in STATEMENTS:
        names, cols = fc.sql_columns(stmt)
        for rank in (0, 1):
            gn, gc = outs[rank][stmt]
            assert gn == names, stmt
            if " join " in stmt:                                  # rows come grouped by owner rank: same multiset
                a = np.stack([np.asarray(x, dtype=np.int64) for x in gc], axis=1)
                e = np.stack([np.asarray(x, dtype=np.int64) for x in cols], axis=1)
                assert np.array_equal(a[np.lexsort(a.T[::-1])], e[np.lexsort(e.T[::-1])]), stmt
            else:
                for x, y in zip(gc, cols):
                    if x.dtype.kind == "f":
                        assert np.allclose(x, y, rtol=1e-6), stmt
                    else:
                        assert np.array_equal(x, y), stmt
    n, G = len(df), 1 << 14
    kk = (np.arange(n, dtype=np.int64) * 2654435761 % G).astype(np.int32)
    keep = df.p.to_numpy() > 0.5
    es = np.bincount(kk[keep], weights=df.v.to_numpy()[keep].astype(np.float64), minlength=G).astype(np.float32)
    ec = np.bincount(kk[keep], minlength=G).astype(np.int64)
    for rank in (0, 1):
        assert np.array_equal(outs[rank]["fgb"][0], es) and np.array_equal(outs[rank]["fgb"][1], ec)
