"""N>1 host logic under gloo, world_size 2, on CPU: row-range sharding, the
all-reduce merge of dense GROUP BY partials and the offset exchange for
order-preserving filters.  Per-shard partials come from the oracle (test
infrastructure); the merge code under test is harkdb_amd.dist."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEED = 0x4861726B4442


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, G, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from harkdb_amd import dist as hd
    from oracle import oracle as ora
    r, _, w = hd.init_process_group("cpu")
    lo, hi = hd.shard_range(n, r, w)
    p, k, v = ora.gen_columns(SEED, lo, hi - lo, G, True)
    _, s64, cnt = ora.filter_groupby_dense_f32(p, k, v, ">", 0.5, G)
    sum_t, cnt_t = torch.from_numpy(s64.copy()), torch.from_numpy(cnt.copy())
    hd.allreduce_partials(sum_t, cnt_t)
    idx = ora.filter_indices(p, ">", 0.5)
    off, total = hd.shard_offsets(len(idx))
    gidx = hd.global_row_index(idx, r, n, w)
    mn, mx = torch.tensor([float(v.min())]), torch.tensor([float(v.max())])
    hd.allreduce_minmax(mn, mx)
    q.put((r, sum_t.numpy(), cnt_t.numpy(), off, total, gidx, (lo, hi), mn.item(), mx.item()))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,G", [(100_003, 1 << 12), (7, 16)])
def test_two_rank_merge_matches_single(oracle, n, G):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, G, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    outs = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    p, k, v = oracle.gen_columns(SEED, 0, n, G, True)
    _, s64, cnt = oracle.filter_groupby_dense_f32(p, k, v, ">", 0.5, G)
    idx = oracle.filter_indices(p, ">", 0.5)
    for r, s, c, off, total, gidx, (lo, hi), mn, mx in outs:
        assert np.array_equal(s, s64) and np.array_equal(c, cnt)        # every rank holds the merged table
        assert total == len(idx)
        assert np.array_equal(idx[off: off + len(gidx)], gidx)          # shard results concatenate in rank order
        assert mn == v.min() and mx == v.max()
    assert outs[0][6][0] == 0 and outs[0][6][1] == outs[1][6][0] and outs[1][6][1] == n


def test_shard_range_properties():
    from harkdb_amd.dist import shard_range
    for n in (0, 1, 5, 1000, 10**9 + 7):
        for world in (1, 2, 4, 8):
            rs = [shard_range(n, r, world) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(rs, rs[1:]))
            assert all(lo % 4 == 0 for lo, hi in rs if lo < n)


def _merge_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    from harkdb_amd import dist as hd
    hd.init_process_group("cpu")
    rng = np.random.default_rng(100 + rank)
    G = 64
    cnt = rng.integers(0, 3, size=G).astype(np.int64) * (rng.random(G) < 0.6)     # many slots are empty on a rank
    junk = lambda a: np.where(cnt > 0, a, rng.integers(-9, 9, size=G).astype(a.dtype))   # empty slots hold anything
    sf = junk(rng.integers(0, 100, size=G).astype(np.float32))
    si = junk(rng.integers(-1000, 1000, size=G).astype(np.int64))
    mn = junk(rng.integers(-50, 50, size=G).astype(np.int32))
    mxu = junk(rng.integers(0, 2**32, size=G, dtype=np.uint64).astype(np.uint32))
    mxf = junk(rng.random(G).astype(np.float32))
    tens = [torch.from_numpy(sf), torch.from_numpy(si), torch.from_numpy(mn), torch.from_numpy(mxu.view(np.int32)), torch.from_numpy(mxf)]
    merged = hd.merge_slot_columns(tens, ["sum", "sum", "min", "max", "max"], [np.float32, np.int64, np.int32, np.uint32, np.float32], torch.from_numpy(cnt))
    keys = np.unique(rng.integers(-50, 50, size=40)).astype(np.int32)
    g = hd.gather_columns([keys])[0]
    gt, counts = hd.gather_tensors([torch.from_numpy(keys), torch.from_numpy(keys.astype(np.float32))])
    q.put((rank, cnt, sf, si, mn, mxu, mxf, [m.numpy() for m in merged], keys, g, [t.numpy() for t in gt], counts))
    import torch.distributed as dist
    dist.barrier(); dist.destroy_process_group()


def test_merge_slot_columns_and_gather_two_ranks():
    """The elementwise merge behind the dense sharded GROUP BY (all-reduce SUM / MIN / MAX of per-slot partials, empty
    slots neutralised, unsigned MAX above 2^31, f32 sums widened) and the rank-order gathers, over gloo."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_merge_worker, args=(r, world, port, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    outs = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    cnts = [o[1] for o in outs]
    total = cnts[0] + cnts[1]
    live = total > 0

    def fold(j, f, neutral, dt):
        return f(*[np.where(o[1] > 0, o[j].astype(dt), neutral) for o in outs])

    for o in outs:
        m = o[7]
        assert np.array_equal(m[5], total)
        assert m[0].dtype == np.float64 and np.array_equal(m[0], fold(2, np.add, 0.0, np.float64))
        assert np.array_equal(m[1], fold(3, np.add, 0, np.int64))
        assert np.array_equal(m[2][live], fold(4, np.minimum, np.iinfo(np.int32).max, np.int32)[live])
        assert np.array_equal(m[3].view(np.uint32)[live], fold(5, np.maximum, 0, np.uint32)[live])
        assert np.array_equal(m[4][live], fold(6, np.maximum, -np.inf, np.float32)[live])
        allk = np.concatenate([x[8] for x in outs])
        assert np.array_equal(o[9], allk)                                        # rank-order concatenation (numpy out)
        assert np.array_equal(o[10][0], allk) and np.array_equal(o[10][1], allk.astype(np.float32))
        assert o[11] == [len(x[8]) for x in outs]


def _exchange_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from harkdb_amd import dist as hd
    hd.init_process_group("cpu")
    rng = np.random.default_rng(7 + rank)
    n = 1000 + 37 * rank
    keys = rng.integers(0, 500, size=n).astype(np.int64)
    vals = rng.random(n).astype(np.float32)
    dest = (keys * 2654435761 % 2**32 * world >> 32).astype(np.int64)       # any deterministic key -> rank map
    order = np.argsort(dest, kind="stable")
    counts = np.bincount(dest, minlength=world).tolist()
    recv, rc = hd.exchange_columns([torch.from_numpy(keys[order]), torch.from_numpy(vals[order])], counts)
    q.put((rank, keys, vals, dest, recv[0].numpy(), recv[1].numpy(), rc))
    import torch.distributed as dist
    dist.barrier(); dist.destroy_process_group()


def test_exchange_columns_all_to_all_two_ranks():
    """The repartition exchange: every row reaches the rank its key hashes to,
    rows from source rank 0 before rows from source rank 1, order inside a source kept."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_exchange_worker, args=(r, world, port, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    outs = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    for me, o in enumerate(outs):
        exp_k = np.concatenate([src[1][src[3] == me] for src in outs])
        exp_v = np.concatenate([src[2][src[3] == me] for src in outs])
        assert np.array_equal(o[4], exp_k) and np.array_equal(o[5], exp_v)
        assert o[6] == [int((src[3] == me).sum()) for src in outs]


def _samplesort_worker(rank, world, port, descending, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from harkdb_amd import dist as hd
    hd.init_process_group("cpu")
    rng = np.random.default_rng(21)
    n = 5003
    keys_all = rng.integers(-50, 50, size=n).astype(np.int32)          # many ties: stability matters
    rowid_all = np.arange(n, dtype=np.int64)
    lo, hi = hd.shard_range(n, rank, world)
    keys, rowid = keys_all[lo:hi], rowid_all[lo:hi]
    # the protocol of ShardedFutharkContext._orderby with numpy standing in for the two device operators
    splitters = hd.gather_splitters(keys[hd.sample_positions(len(keys), 64)], world)
    d = np.searchsorted(splitters, keys, side="right")                 # = hark_op_partition_by_range
    dest = (len(splitters) - d) if descending else d
    order = np.argsort(dest, kind="stable")
    counts = np.bincount(dest, minlength=world).tolist()
    recv, rc = hd.exchange_columns([torch.from_numpy(keys[order]), torch.from_numpy(rowid[order])], counts)
    rk, rr = recv[0].numpy(), recv[1].numpy()
    perm = np.argsort(-rk.astype(np.int64) if descending else rk, kind="stable")     # = the local stable radix sort
    cols = hd.gather_columns([rk[perm], rr[perm]])
    q.put((rank, splitters, cols[0], cols[1]))
    import torch.distributed as dist
    dist.barrier(); dist.destroy_process_group()


@pytest.mark.parametrize("descending", [False, True])
def test_sample_sort_exchange_two_ranks(descending):
    """Distributed ORDER BY: pooled samples give every rank the same splitters, the range
    exchange + local stable sort + rank-order concatenation equals one global stable sort."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_samplesort_worker, args=(r, world, port, descending, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    outs = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    rng = np.random.default_rng(21)
    keys_all = rng.integers(-50, 50, size=5003).astype(np.int32)
    exp = np.argsort(-keys_all.astype(np.int64) if descending else keys_all, kind="stable")
    assert np.array_equal(outs[0][1], outs[1][1]) and len(outs[0][1]) == world - 1
    for o in outs:
        assert np.array_equal(o[3], exp) and np.array_equal(o[2], keys_all[exp])


def test_choose_splitters_properties():
    from harkdb_amd import dist as hd
    a = np.arange(1000, dtype=np.float32)[::-1]
    s = hd.choose_splitters(a, 8)
    assert len(s) == 7 and np.all(np.diff(s) > 0) and s.dtype == np.float32
    assert len(hd.choose_splitters(a, 1)) == 0 and len(hd.choose_splitters(a[:0], 4)) == 0
    assert hd.sample_positions(10, 64).tolist() == list(range(10)) and len(hd.sample_positions(10**6, 64)) == 64
    assert hd.sample_positions(0).size == 0


# ---- round 2: tensor gather (no pickling), reduce-scatter + all-gather knob, pipelined steps -------------------------
def _gather_worker(rank, world, port, mode, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HARK_GATHER=mode)
    from harkdb_amd import dist as hd
    hd.init_process_group("cpu")
    rng = np.random.default_rng(5 + rank)
    out = []
    for n in ((0, 7)[rank], 1000 + 333 * rank, 0):                        # an empty shard, ragged shards, all empty
        cols = [rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32), rng.integers(-2**62, 2**62, n).astype(np.int64),
                rng.random(n).astype(np.float32), torch.from_numpy(rng.integers(-9, 9, n).astype(np.int32))]
        got = hd.gather_columns(cols)
        out.append(([c.numpy() if isinstance(c, torch.Tensor) else c for c in cols], got))
    q.put((rank, out))
    import torch.distributed as dist
    dist.barrier(); dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["tensor", "object"])
def test_gather_columns_ragged_mixed_dtypes(mode):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gather_worker, args=(r, world, port, mode, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    outs = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    for case in range(3):
        for j in range(4):
            exp = np.concatenate([outs[r][1][case][0][j] for r in range(world)])
            for r in range(world):
                got = outs[r][1][case][1][j]
                assert got.dtype == exp.dtype and np.array_equal(got, exp)          # rank-order concatenation, bit-exact, dtype kept


class _FakePlan:
    """Stands in for FgbPlan on CPU: 'run' accumulates numpy partials into torch accumulators."""

    def __init__(self, G):
        self.G, self.sum_t, self.cnt_t = G, torch.zeros(G, dtype=torch.float64), torch.zeros(G, dtype=torch.int64)
        self.finished = []

    def reset(self):
        self.sum_t.zero_(); self.cnt_t.zero_()

    def run(self, p, cmp, thr, k, v, n):
        keep = p > thr
        self.sum_t += torch.from_numpy(np.bincount(k[keep], weights=v[keep], minlength=self.G))
        self.cnt_t += torch.from_numpy(np.bincount(k[keep], minlength=self.G))

    def finish(self, sum_out, count_out, check=True):
        sum_out[:] = self.sum_t.numpy(); count_out[:] = self.cnt_t.numpy()

    def check(self):
        self.checked = True


def _pipeline_worker(rank, world, port, how, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HARK_ALLREDUCE=how)
    from harkdb_amd import dist as hd
    hd.init_process_group("cpu")
    G, steps = 64, 5
    plans = [_FakePlan(G), _FakePlan(G)]
    job = hd.ShardedFgb(None, plans[0], "cpu", plan2=plans[1], acc_tensors=[(pl.sum_t, pl.cnt_t) for pl in plans])
    assert job.pipelined
    outs = [(np.zeros(G), np.zeros(G, dtype=np.int64)) for _ in range(steps)]
    for i in range(steps):
        rng = np.random.default_rng(1000 * i + rank)
        n = 500
        job.step(rng.random(n), ">", 0.5, rng.integers(0, G, n), rng.integers(0, 16, n).astype(np.float64), n, outs[i][0], outs[i][1])
    job.flush()
    q.put((rank, outs))
    import torch.distributed as dist
    dist.barrier(); dist.destroy_process_group()


@pytest.mark.parametrize("how", ["allreduce", "rs_ag"])
def test_pipelined_steps_and_allreduce_variants(how):
    """ShardedFgb with two plans: step i's all-reduce is awaited when step i+1 is issued (or by flush); every step's
    outputs equal the sum over ranks of that step's partials, for the plain all-reduce and for reduce-scatter + all-gather."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_pipeline_worker, args=(r, world, port, how, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    outs = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    G = 64
    for i in range(5):
        es, ec = np.zeros(G), np.zeros(G, dtype=np.int64)
        for rank in range(world):
            rng = np.random.default_rng(1000 * i + rank)
            p, k, v = rng.random(500), rng.integers(0, G, 500), rng.integers(0, 16, 500).astype(np.float64)
            keep = p > 0.5
            es += np.bincount(k[keep], weights=v[keep], minlength=G); ec += np.bincount(k[keep], minlength=G)
        for rank in range(world):
            assert np.array_equal(outs[rank][1][i][0], es) and np.array_equal(outs[rank][1][i][1], ec)


@pytest.mark.parametrize("kind", ["txt", "csv", "frame", "ndarray"])
@pytest.mark.parametrize("world", [2, 3, 8])
def test_shards_of_a_host_table_keep_the_whole_tables_schema(tmp_path, kind, world):
    """ShardedFutharkContext.create_table slices host sources per rank: every shard must carry the schema of the unsharded
    table (a TXT table's c1, c2, ... were renamed col1, col2, ... by the ndarray slice: ADVICE r03) and the shards must
    concatenate to the table, in rank order."""
    import pandas as pd
    from harkdb_amd import dist as hd
    from harkdb_amd.table import Table
    rng = np.random.default_rng(world)
    mat = rng.integers(-50, 50, size=(1003, 3))
    if kind == "txt":
        src = str(tmp_path / "x.txt")
        np.savetxt(src, mat, fmt="%d")
    elif kind == "csv":
        src = str(tmp_path / "x.csv")
        pd.DataFrame(mat, columns=["a", "b", "c"]).to_csv(src, index=False)
    elif kind == "frame":
        src = pd.DataFrame(mat, columns=["a", "b", "c"])
    else:
        src = mat
    whole = Table("t", src)
    parts = [hd.shard_of_host_table("t", src, r, world) for r in range(world)]
    for part in parts:
        assert part.get_schema() == whole.get_schema()
    got = np.concatenate([np.asarray(part.get_data(), dtype=np.float64).reshape(-1, 3) for part in parts])
    assert np.array_equal(got, np.asarray(whole.get_data(), dtype=np.float64))
    if kind == "txt":
        assert whole.get_schema() == ["c1", "c2", "c3"]
