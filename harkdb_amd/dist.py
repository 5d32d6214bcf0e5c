"""Row-range sharding of one table over the GPUs of a node.

One process per GPU (torch.distributed; backend "nccl" IS RCCL on ROCm, "gloo"
on CPU).  Rank r owns the contiguous row range shard_range(n, r, world) of
every column, so:
  * projection / WHERE need no data-path collective: global row index =
    shard base + local index and results concatenate in rank order (one
    all_gather of the per-shard survivor counts gives the output offsets);
  * GROUP BY over a dense key domain needs ONE exchange: every rank builds a
    partial (sum f64, count i64) table of G slots with the fused kernel and the
    partials are summed with an all-reduce (16 B x G: 16 MiB for G = 2^20).

The functions here are device-agnostic (they only see torch tensors), which
is what lets the N>1 logic run under gloo on CPU in tests.
"""
import os

import numpy as np


def shard_range(n, rank, world):
    """Rows [lo, hi) of rank `rank`: contiguous, balanced, 4-row aligned so the
    16-byte loads of the kernels stay aligned inside a shard."""
    per = -(-n // world)
    per = -(-per // 4) * 4
    lo = min(n, rank * per)
    return lo, min(n, lo + per)


def init_process_group(device_type=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun contract)."""
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if device_type is None:
        device_type = "cuda" if torch.cuda.is_available() else "cpu"
    if (world > 1 or "RANK" in os.environ) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        backend = os.environ.get("HARK_DIST_BACKEND") or ("nccl" if device_type == "cuda" else "gloo")
        if device_type == "cuda":
            torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            # gloo with device tensors: a TEST arrangement (several ranks on one GPU, which RCCL refuses);
            # collectives on device tensors are staged through the host by _host_staged below
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local, world


def _host_staged(t):
    """(tensor to hand to the collective, copy-back function).  RCCL takes device tensors as they are; under
    gloo (tests: two ranks sharing one GPU) device tensors travel through the host."""
    import torch.distributed as dist
    if t.is_cuda and dist.get_backend() == "gloo":
        h = t.cpu()
        return h, lambda: t.copy_(h)
    return t, lambda: None


def share_stream(eng, device):
    """Put the engine's kernels and torch (RCCL collectives, tensor ops, events) on ONE
    stream: a torch side stream made current.  torch's default stream has handle 0, which
    hark_context_set_stream reads as "the context's own stream" -- kernels there would not be
    ordered against collectives, so the default stream is never shared."""
    import torch
    st = torch.cuda.Stream(device=device)
    torch.cuda.set_stream(st)
    assert st.cuda_stream != 0
    eng.set_stream(st.cuda_stream)
    return st


class _DevicePtr:
    """Exposes a raw device address to torch without a copy."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (int(ptr), False), "version": 3, "strides": None}


def tensor_from_ptr(ptr, n, dtype, device):
    """Zero-copy torch view of `n` elements at device address `ptr`."""
    import torch
    typestr = {"float64": "<f8", "int64": "<i8", "float32": "<f4", "int32": "<i4", "uint32": "<u4"}[np.dtype(dtype).name]
    return torch.as_tensor(_DevicePtr(ptr, n, typestr), device=device)


def _allreduce_sum(t, group, how, async_op):
    """SUM over ranks of t, in place.  how = "allreduce": one ncclAllReduce (RCCL picks ring / tree by size);
    "rs_ag": reduce-scatter of the G / world slices, then all-gather of the reduced slices -- the two halves of a ring
    all-reduce issued separately, so that on xGMI (point-to-point, 7 links per GPU) both can be A/B-ed on hardware
    (HARK_ALLREDUCE=rs_ag; needs len(t) divisible by the world size, else falls back)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if how == "rs_ag" and world > 1 and t.numel() % world == 0:
        # both halves are issued asynchronously: a process group runs its collectives in issue order on its own stream,
        # so the all-gather starts when the reduce-scatter has filled `part`, and the caller's stream waits for neither
        part = torch.empty(t.numel() // world, dtype=t.dtype, device=t.device)
        w1 = dist.reduce_scatter_tensor(part, t, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        w2 = dist.all_gather_into_tensor(t, part, group=group, async_op=async_op)
        return _ChainedWork([w1, w2], keep=part) if async_op else None
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


class _ChainedWork:
    """Several pending collectives that finish in order, awaited as one (and the scratch tensor they share)."""

    def __init__(self, works, keep=None):
        self.works, self.keep = [w for w in works if w is not None], keep

    def wait(self):
        for w in self.works:
            w.wait()
        return True


def allreduce_partials(sum_t, cnt_t, group=None, wait=True):
    """Merge per-shard dense GROUP BY partials in place: SUM over ranks of the
    f64 sums and of the i64 counts (the RCCL all-reduce of SURVEY.md 8(e)).  wait=False returns the pending
    work handles (call .wait() on each before reading the tensors): the collectives then run on RCCL's own
    stream beside whatever the caller enqueues next (ShardedFgb pipelines the next step's kernels under them)."""
    import torch.distributed as dist
    works = []
    if dist.is_initialized():          # also with one rank: keeps the single-GPU run on the same code path
        how = os.environ.get("HARK_ALLREDUCE", "allreduce")
        if sum_t.is_cuda and dist.get_backend() == "gloo":               # tests: ranks sharing one GPU
            for t in (sum_t, cnt_t):
                x, back = _host_staged(t)
                dist.all_reduce(x, op=dist.ReduceOp.SUM, group=group)
                back()
        else:
            # both collectives are issued before either is awaited: the second one's launch overlaps the first
            works = [_allreduce_sum(t, group, how, True) for t in (sum_t, cnt_t)]
            if wait:
                for w in works:
                    w.wait()
                works = []
    return (sum_t, cnt_t) if wait else works


def allreduce_minmax(min_t, max_t, group=None):
    """MIN / MAX partials (ncclMin / ncclMax)."""
    import torch.distributed as dist
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        for t, op in ((min_t, dist.ReduceOp.MIN), (max_t, dist.ReduceOp.MAX)):
            x, back = _host_staged(t)
            dist.all_reduce(x, op=op, group=group)
            back()
    return min_t, max_t


def shard_offsets(local_count, device="cpu", group=None):
    """Output offset of this rank's rows after an order-preserving per-shard
    filter: exclusive prefix over ranks of the survivor counts.  Returns
    (offset of this rank, total)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return 0, int(local_count)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if dist.get_backend() == "gloo":
        device = "cpu"
    mine = torch.tensor([int(local_count)], dtype=torch.int64, device=device)
    allc = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allc, mine, group=group)
    counts = [int(c.item()) for c in allc]
    return sum(counts[:rank]), sum(counts)


def global_row_index(local_index, rank, n, world):
    """Local compaction indices -> indices into the unsharded table."""
    return local_index + shard_range(n, rank, world)[0]


class ShardedFgb:
    """SELECT k, SUM(v), COUNT(*) WHERE p <cmp> thr GROUP BY k over a row-range
    shard per rank: local fused kernel -> all-reduce of the accumulators ->
    finish.  `eng`/`plan` are this rank's Engine and FgbPlan.

    With more than one rank and a second plan (`plan2`) steps are PIPELINED: the all-reduce of step i runs on RCCL's
    stream while the kernels of step i+1 (which accumulate into the other plan) run on ours; step i is finished --
    results written to its outputs -- when step i+1 is issued, or by flush().  HARK_OVERLAP=0 switches it off."""

    def __init__(self, eng, plan, device, plan2=None, acc_tensors=None):
        self.eng, self.plan, self.device = eng, plan, device
        self.plans = [plan] + ([plan2] if plan2 is not None and os.environ.get("HARK_OVERLAP", "1") != "0" else [])
        self.acc = []
        for j, pl in enumerate(self.plans):
            if acc_tensors is not None:                                   # tests: accumulators that are not HBM addresses
                self.acc.append(acc_tensors[j])
                continue
            s_ptr, c_ptr = pl.acc_ptrs()
            self.acc.append((tensor_from_ptr(s_ptr, pl.G, np.float64, device), tensor_from_ptr(c_ptr, pl.G, np.int64, device)))
        self.sum_t, self.cnt_t = self.acc[0]
        self.pending = None
        self.turn = 0

    @property
    def pipelined(self):
        return len(self.plans) > 1

    def step(self, p, cmp, thr, k, v, n, sum_out=None, count_out=None, check=None):
        """One query over this rank's shard.  Serial mode (one plan): the outputs are written when step() returns, and
        with check=True (the default there) the device error word -- a surviving row whose key lies outside [0, G) -- is
        read and raised right away.  A caller that loops (bench.py) passes check=False to keep the host out of the loop
        and MUST call flush() before trusting the outputs.  Pipelined mode (two plans): sum_out / count_out of step i are
        only written when step i + 1 is issued or by flush(), and errors are only raised by flush()."""
        if not self.pipelined:
            self.plan.reset()
            self.plan.run(p, cmp, thr, k, v, n)
            allreduce_partials(self.sum_t, self.cnt_t)
            self.plan.finish(sum_out, count_out, check=True if check is None else bool(check))
            return
        i = self.turn
        self.turn = 1 - i
        plan, (sum_t, cnt_t) = self.plans[i], self.acc[i]
        plan.reset()
        plan.run(p, cmp, thr, k, v, n)
        works = allreduce_partials(sum_t, cnt_t, wait=False)         # runs beside the NEXT step's kernels
        prev, self.pending = self.pending, (plan, works, sum_out, count_out)
        self._finish(prev)

    def _finish(self, item):
        if item is None:
            return
        plan, works, sum_out, count_out = item
        for w in works:
            w.wait()                                                  # our stream waits for the collective, the host does not
        plan.finish(sum_out, count_out, check=False)

    def flush(self):
        """Finish the step still in flight (its all-reduce has been running beside nothing since the last step())."""
        item, self.pending = self.pending, None
        self._finish(item)
        for pl in self.plans:                                         # the sticky error words of the steps since the last flush
            pl.check()


# ---------------------------------------------------------------------------
# Sharded SQL surface: FutharkContext over row-range shards
# ---------------------------------------------------------------------------
def gather_columns(cols, group=None, device=None, np_dtypes=None):
    """Concatenate per-rank result columns in rank order on every rank (row order = the unsharded table's order).

    One all-gather of the row counts, then ONE `all_gather_into_tensor` of a byte buffer that holds every column padded
    to the longest shard -- device tensors travel over RCCL as they are (no pickling, no host copy before the
    collective); `cols` may be numpy arrays (uploaded to `device` first when it is a GPU) or 1-D torch tensors.
    HARK_GATHER=object selects the old `all_gather_object` path (A/B on hardware).  Returns numpy arrays.
    `np_dtypes` (one per column) names the dtypes of the returned arrays; without it they are derived from the local
    tensors, which cannot tell uint32 from int32 (torch carries u32 bit patterns in int32 tensors) -- a rank with an
    empty shard would otherwise answer with another dtype than its peers."""
    import torch
    import torch.distributed as dist
    def retype(arrs):
        return arrs if np_dtypes is None else [a.view(np.dtype(d)) if a.dtype != np.dtype(d) else a for a, d in zip(arrs, np_dtypes)]
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return retype([c.cpu().numpy() if isinstance(c, torch.Tensor) else c for c in cols])
    world = dist.get_world_size(group)
    if os.environ.get("HARK_GATHER", "tensor") == "object":
        host = retype([c.cpu().numpy() if isinstance(c, torch.Tensor) else c for c in cols])
        parts = [None] * world
        dist.all_gather_object(parts, host, group=group)
        return [np.concatenate([p[j] for p in parts]) for j in range(len(host))]
    staged = dist.get_backend() == "gloo"                                 # CPU tests, or ranks sharing one GPU
    if not staged and device is None:
        device = torch.device("cuda", torch.cuda.current_device())        # RCCL moves device memory only
    dev = torch.device("cpu") if staged else torch.device(device)
    given, np_dtypes, tens = np_dtypes, [], []
    for c in cols:
        if isinstance(c, torch.Tensor):
            np_dtypes.append(np.dtype(str(c.dtype).replace("torch.", "")))
            tens.append(c.to(dev).contiguous())
        else:
            a = np.ascontiguousarray(c)
            np_dtypes.append(a.dtype)
            tens.append(torch.from_numpy(a.view(np.int32) if a.dtype == np.uint32 else a).to(dev))
    if given is not None:
        np_dtypes = [np.dtype(d) for d in given]
    n_local = int(tens[0].numel()) if tens else 0
    mine = torch.tensor([n_local], dtype=torch.int64, device=dev)
    allc = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(allc, mine, group=group)
    counts = [int(x) for x in allc.tolist()]
    maxn = max(counts)
    if maxn == 0 or not tens:
        return [np.empty(0, dtype=dt) for dt in np_dtypes]
    widths = [t.element_size() for t in tens]
    row_bytes = sum(widths)
    send = torch.zeros(maxn * row_bytes, dtype=torch.uint8, device=dev)   # column-major: column j occupies maxn * width_j bytes
    off = 0
    for t, w in zip(tens, widths):
        if n_local:
            send[off: off + n_local * w] = t.reshape(-1).view(torch.uint8)
        off += maxn * w
    recv = torch.empty(world * maxn * row_bytes, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(recv, send, group=group)
    host = recv.cpu().numpy()
    out, off = [], 0
    for dt, w in zip(np_dtypes, widths):
        pieces = [host[r * maxn * row_bytes + off: r * maxn * row_bytes + off + counts[r] * w] for r in range(world)]
        out.append(np.concatenate(pieces).view(dt))
        off += maxn * w
    return out


def result_tensors(res, device, limit=None):
    """The columns of a device-resident Result as zero-copy torch tensors (the first `limit` rows)."""
    n, m = res.shape
    if limit is not None:
        n = min(n, max(int(limit), 0))
    return [tensor_from_ptr(res.device_ptr(j), n, res.dtype(j), device) if n else __import__("torch").empty(0, dtype=getattr(__import__("torch"), _TORCH_OF[np.dtype(res.dtype(j)).name]), device=device)
            for j in range(m)], [np.dtype(res.dtype(j)) for j in range(m)]


_MERGE = {"sum": np.add, "count": np.add, "min": np.minimum, "max": np.maximum, "prod": np.multiply}


def merge_grouped(keys, aggs, funcs, group=None):
    """Merge per-rank GROUP BY results over sparse keys: gather (key, partial
    aggregate) rows from every rank and fold equal keys.  `funcs[j]` names how
    partials of aggregate j combine ("sum", "count", "min", "max", "prod");
    AVG must be carried as a (sum, count) pair by the caller.  Returns keys
    ascending."""
    keys, aggs = gather_columns([keys], group)[0], gather_columns(list(aggs), group)
    order = np.argsort(keys, kind="stable")
    keys, aggs = keys[order], [a[order] for a in aggs]
    heads = np.ones(len(keys), dtype=bool)
    heads[1:] = keys[1:] != keys[:-1]
    starts = np.flatnonzero(heads)
    out = [_MERGE[f].reduceat(a, starts) if len(a) else a for a, f in zip(aggs, funcs)]
    return keys[heads], out


# ---------------------------------------------------------------------------
# Repartition by key hash: the all-to-all of SURVEY.md 8(e) (join, sparse GROUP BY)
# ---------------------------------------------------------------------------
def exchange_columns(send_cols, send_counts, group=None):
    """All-to-all of row-partitioned columns.  Every tensor of `send_cols` holds
    this rank's rows GROUPED BY DESTINATION RANK (send_counts[r] rows for rank
    r, in rank order).  Returns (received tensors, received counts per source)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        return list(send_cols), list(send_counts)
    dev = send_cols[0].device if send_cols else "cpu"
    staged = dist.get_backend() == "gloo" and torch.device(dev).type == "cuda"      # tests: ranks sharing one GPU
    cdev = "cpu" if staged else dev
    cnt = torch.tensor(list(send_counts), dtype=torch.int64, device=cdev)
    rc = torch.empty_like(cnt)
    dist.all_to_all_single(rc, cnt, group=group)
    recv_counts = [int(x) for x in rc.tolist()]
    out = []
    for c in send_cols:
        src = c.contiguous().cpu() if staged else c.contiguous()
        r = torch.empty(sum(recv_counts), dtype=c.dtype, device=cdev)
        dist.all_to_all_single(r, src, output_split_sizes=recv_counts, input_split_sizes=list(send_counts), group=group)
        out.append(r.to(dev) if staged else r)
    return out, recv_counts


_TORCH_OF = {"int32": "int32", "uint32": "int32", "float32": "float32", "int64": "int64"}     # uint32 travels as its bit pattern


def repartition_device(eng, ptrs, dtypes, n, key_index, device, world, group=None, splitters=None, descending=False):
    """Partition n rows held in device columns (raw pointers `ptrs`) by column
    `key_index` on the GPU and all-to-all them: by key hash
    (hark_op_partition_by_hash) or, when `splitters` is given, by key range
    (hark_op_partition_by_range: rank r receives the r-th range, rows of one
    source in table order).  Returns (torch tensors owning the received columns, rows)."""
    import torch
    perm = torch.empty(max(n, 1), dtype=torch.int32, device=device)
    if splitters is None:
        counts = eng.partition_by_hash(ptrs[key_index], dtypes[key_index], n, world, perm.data_ptr())
    else:
        counts = eng.partition_by_range(ptrs[key_index], dtypes[key_index], n, splitters, descending, perm.data_ptr())
    send = []
    for ptr, dt in zip(ptrs, dtypes):
        buf = torch.empty(max(n, 1), dtype=getattr(torch, _TORCH_OF[np.dtype(dt).name]), device=device)[:n]
        eng.gather(ptr, dt, perm.data_ptr(), buf.data_ptr(), n)
        send.append(buf)
    recv, rc = exchange_columns(send, counts, group)
    return recv, sum(rc)


# ---------------------------------------------------------------------------
# Sample sort: the exchange of SURVEY.md 8(e) "SORT BY"
# ---------------------------------------------------------------------------
SAMPLES_PER_RANK = 256


def sample_positions(n, samples=SAMPLES_PER_RANK):
    """Evenly spaced row positions of a shard to sample its sort keys at."""
    m = min(int(n), int(samples))
    return (np.arange(m, dtype=np.int64) * int(n)) // max(m, 1)


def choose_splitters(all_samples, world):
    """world-1 ascending splitters at the quantiles of the pooled samples."""
    a = np.sort(np.asarray(all_samples), kind="stable")
    if world <= 1 or a.size == 0:
        return a[:0]
    return a[(np.arange(1, world) * a.size) // world]


def gather_splitters(local_sample, world, group=None):
    """All ranks pool their key samples (all-gather) and derive the SAME splitters."""
    import torch.distributed as dist
    local_sample = np.asarray(local_sample)
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return choose_splitters(local_sample, world)
    parts = [None] * dist.get_world_size(group)
    dist.all_gather_object(parts, local_sample, group=group)
    return choose_splitters(np.concatenate([p.astype(local_sample.dtype, copy=False) for p in parts]), world)


_SECOND_LEVEL = {"sum": "sum", "count": "sum", "min": "min", "max": "max", "prod": "prod"}


class ShardedFutharkContext:
    """FutharkContext whose tables are row-range shards, one rank per GPU.

    create_table() takes the FULL host table on every rank and keeps only this
    rank's rows (shard_range); sql() runs the local operators on the shard and
    merges: projection / WHERE results concatenate in rank order, GROUP BY
    partials are all-reduced (dense keys, device side) or merged by key."""

    def __init__(self, device=None, device_exchange=None):
        import torch
        from .context import FutharkContext
        self.rank, local, self.world = init_process_group()
        self.device = torch.device("cuda", local if device is None else device)
        torch.cuda.set_device(self.device)
        self.local = FutharkContext(device=self.device.index, sql_mode=True)
        # kernels, gathers and RCCL collectives are ordered by ONE stream
        self.stream = share_stream(self.local.FutEnv, self.device)
        self.rows = {}
        # sparse GROUP BY / JOIN exchange rows with an RCCL all-to-all when there is more than one rank
        self.device_exchange = (self.world > 1) if device_exchange is None else bool(device_exchange)

    def create_table(self, table_name, table):
        from .table import Table
        full = Table(table_name, table)
        n = full.get_data().shape[0]
        lo, hi = shard_range(n, self.rank, self.world)
        cols = [c[lo:hi] for c in full.host_columns()]
        import pandas as pd
        self.local.create_table(table_name, pd.DataFrame({h: c for h, c in zip(full.get_schema(), cols)}))
        self.rows[table_name] = (n, lo, hi)

    def drop_table(self, table_name):
        self.local.drop_table(table_name)
        self.rows.pop(table_name)

    def sql(self, sql_statement):
        names, cols = self.sql_columns(sql_statement)
        dtype = np.result_type(*[c.dtype for c in cols]) if cols else np.int32
        out = np.empty((len(cols[0]) if cols else 0, len(cols)), dtype=dtype)
        for j, c in enumerate(cols):
            out[:, j] = c
        return out

    def sql_columns(self, sql_statement):
        from .parse import sql_parse
        ir = sql_parse(self.local.tables, sql_statement)
        if ir.get("orderby_all"):
            raise Exception("ORDER BY on several keys over sharded tables is not built yet")
        if ir.get("join"):
            return self._join(ir)
        if "groupbys" not in ir:
            if "orderby" in ir and (self.world > 1 or self.device_exchange):
                return self._orderby(ir)
            limit = ir.pop("limit", None)
            stmt = sql_statement if limit is None else sql_statement[: sql_statement.lower().rindex("limit")]
            names, res = self.local.select_result(stmt)                 # device-resident: gathered over RCCL as it is
            tens, dts = result_tensors(res, self.device, limit)         # no rank contributes more than LIMIT rows
            cols = gather_columns(tens, device=self.device, np_dtypes=dts)
            return names, ([c[:limit] for c in cols] if limit is not None else cols)
        return self._groupby(ir)

    def _groupby(self, ir):
        """Local typed GROUP BY on the shard (AVG split into SUM and COUNT),
        merge by key, then HAVING / ORDER BY / LIMIT on the merged G rows."""
        schema = self.local.tables[ir["table_name"]].get_schema()
        table_name, key_name, decode = ir["table_name"], schema[ir["g_col"]], None
        g_cols = ir.get("g_cols", [ir["g_col"]])
        if len(g_cols) > 1:
            # several keys: one composite key column per shard, encoded with the ranges over ALL shards so that every
            # rank encodes alike; the single-key machinery below then runs on a temporary table with that column
            eng, tab = self.local.FutEnv, self.local.tables[ir["table_name"]]
            dev = tab._device
            mine = [eng.column_range(dev, c) for c in g_cols] if dev.shape[0] > 0 else None
            parts = [mine]
            import torch.distributed as dist
            if dist.is_initialized() and dist.get_world_size() > 1:
                parts = [None] * dist.get_world_size()
                dist.all_gather_object(parts, mine)
            parts = [p for p in parts if p is not None]
            mins = [min(p[j][0] for p in parts) for j in range(len(g_cols))] if parts else [0] * len(g_cols)
            spans = [max(p[j][1] for p in parts) - mins[j] + 1 for j in range(len(g_cols))] if parts else [1] * len(g_cols)
            buf, cdt, _, _ = eng.composite_key(dev, g_cols, ranges=(mins, spans))
            m = dev.shape[1]
            table_name, key_name = f"__mk_{ir['table_name']}", "__key"
            self.local.create_table_from_device(table_name, list(schema) + [key_name], [dev.device_ptr(j) for j in range(m)] + [buf.ptr or 0],
                                                [dev.dtype(j) for j in range(m)] + [cdt], dev.shape[0], keepalive=(dev, buf))
            decode = (mins, spans, [dev.dtype(c) for c in g_cols])
        try:
            return self._groupby_merge(ir, schema, table_name, key_name, g_cols, decode)
        finally:
            if decode is not None:
                self.local.drop_table(table_name)

    def _groupby_merge(self, ir, schema, table_name, key_name, g_cols, decode):
        specs = []                       # partial aggregates to compute locally: (func, col)

        def slot(spec):
            if spec[0] == "key":
                return ("key", spec[1])
            if spec[0] == "col":
                raise Exception(f"{schema[spec[1]]} is not an aggregation function or the columns thats grouped on")
            if spec[0] == "avg":
                return ("avg", slot(("sum", spec[1]))[1], slot(("count", None))[1])
            if spec not in specs:
                specs.append(spec)
            return ("agg", specs.index(spec))

        items = [slot(i) for i in ir["items"]]
        having = [(slot(s), cmp, v) for s, cmp, v in ir.get("having", [])]
        order = (slot(ir["orderby"][0]), ir["orderby"][1]) if "orderby" in ir else None
        sel = ", ".join([key_name] + [f"{f}({'*' if c is None else schema[c]})" for f, c in specs])
        where = " and ".join(f"{schema[c]} {cmp} {v!r}" for c, cmp, v in ir.get("where", []))
        stmt = f"select {sel} from {table_name}" + (f" where {where}" if where else "") + f" group by {key_name}"
        if self.device_exchange:
            cols = self._groupby_exchange(stmt, specs)
            order_k = np.argsort(cols[0], kind="stable")
            keys, merged = cols[0][order_k], [c[order_k] for c in cols[1:]]      # every key lives on exactly one rank now
        else:
            _, cols = self.local.sql_columns(stmt)
            widen = [c.astype(np.float64) if c.dtype == np.float32 else c for c in cols[1:]]     # merge f32 partial sums in f64
            keys, merged = merge_grouped(cols[0], widen, [f for f, _ in specs])

        key_cols = {g_cols[0]: keys}
        if decode is not None:                                           # composite -> the key columns (last key = least significant digit)
            comp, key_cols = keys.astype(np.int64), {}
            for c, mn, sp, dt in reversed(list(zip(g_cols, *decode))):
                key_cols[c] = (comp % sp + mn).astype(dt)
                comp = comp // sp

        def value(s):
            if s[0] == "key":
                return key_cols[s[1]]
            if s[0] == "agg":
                f, c = specs[s[1]]
                return merged[s[1]].astype(np.float32) if cols[1 + s[1]].dtype == np.float32 else merged[s[1]]
            return (merged[s[1]] / merged[s[2]]).astype(np.float32)

        keep = np.ones(len(keys), dtype=bool)
        ops = {">": np.greater, ">=": np.greater_equal, "<": np.less, "<=": np.less_equal, "=": np.equal, "!=": np.not_equal}
        for s, cmp, v in having:
            keep &= ops[cmp](value(s), v)
        out = [value(s)[keep] for s in items]
        if order is not None:
            ov = value(order[0])[keep]
            perm = np.argsort(-ov if order[1] else ov, kind="stable") if ov.dtype.kind != "u" else \
                np.argsort((ov.max(initial=0) - ov) if order[1] else ov, kind="stable")
            out = [c[perm] for c in out]
        if "limit" in ir:
            out = [c[: ir["limit"]] for c in out]
        names = [schema[c] if f == "key" else f"{f}({'*' if c is None else schema[c]})" for f, c in ir["items"]]
        return names, out

    # ---- RCCL all-to-all paths ------------------------------------------------------
    def _orderby(self, ir):
        """SELECT ... [WHERE ...] ORDER BY col [DESC] [LIMIT n] over row-range shards as a
        sample sort: local WHERE, pooled key samples -> splitters, range partition on the
        GPU, all-to-all, local stable radix sort, ranges concatenated in rank order.  Ties
        keep table order: a range receives its rows by (source rank, local position)."""
        import torch
        eng, loc = self.local.FutEnv, self.local
        tab = loc.tables[ir["table_name"]]
        dev, schema = tab._device, tab.get_schema()
        okey, desc = ir["orderby"][0][1], bool(ir["orderby"][1])
        need = list(dict.fromkeys([okey] + list(ir["select"])))
        if ir.get("where"):
            cur, cmap = loc._filtered(dev, ir["where"], set(need))
        else:
            cur, cmap = dev, {c: c for c in need}
        n = cur.shape[0]
        ptrs, dts = [cur.device_ptr(cmap[c]) for c in need], [cur.dtype(cmap[c]) for c in need]
        pos = sample_positions(n)
        sample = np.empty(0, dtype=np.dtype(dts[0]))
        if pos.size:
            idx = torch.as_tensor(pos.astype(np.int32), device=self.device)
            buf = torch.empty(pos.size, dtype=getattr(torch, _TORCH_OF[np.dtype(dts[0]).name]), device=self.device)
            eng.gather(ptrs[0], dts[0], idx.data_ptr(), buf.data_ptr(), pos.size)
            sample = buf.cpu().numpy().view(np.dtype(dts[0]))
        splitters = gather_splitters(sample, self.world)
        recv, nrecv = repartition_device(eng, ptrs, dts, n, 0, self.device, self.world, splitters=splitters, descending=desc)
        t = eng.table_from_device(nrecv, [c.data_ptr() for c in recv], dts, keepalive=(recv, cur))
        res = eng.sort(t, 0, [need.index(c) for c in ir["select"]], descending=desc)
        tens, dts = result_tensors(res, self.device, ir.get("limit"))
        cols = gather_columns(tens, device=self.device, np_dtypes=dts)
        if "limit" in ir:
            cols = [c[: ir["limit"]] for c in cols]
        return [schema[c] for c in ir["select"]], cols

    def _result_as_table(self, res):
        eng = self.local.FutEnv
        n, m = res.shape
        return eng.table_from_device(n, [res.device_ptr(j) for j in range(m)], [res.dtype(j) for j in range(m)], keepalive=res)

    def _groupby_exchange(self, stmt, specs):
        """Local partial aggregates -> all-to-all by hash(key) -> second-level
        aggregation of the partials on the owner rank -> gather of the owners' rows."""
        eng = self.local.FutEnv
        names, res = self.local.sql_result(stmt)                    # device-resident [key, partial...]
        n, m = res.shape
        dts = [res.dtype(j) for j in range(m)]
        recv, nrecv = repartition_device(eng, [res.device_ptr(j) for j in range(m)], dts, n, 0, self.device, self.world)
        t = eng.table_from_device(nrecv, [c.data_ptr() for c in recv], dts, keepalive=recv)
        res2 = eng.filter_groupby(t, None, 0, [(_SECOND_LEVEL[f], 1 + j) for j, (f, _) in enumerate(specs)])
        tens, dts = result_tensors(res2, self.device)
        return gather_columns(tens, device=self.device, np_dtypes=dts)

    def _join(self, ir):
        """Both sides are hash-partitioned by the join key and exchanged; every rank
        joins what it owns.  Row order follows (owner rank, key, left row, right row)."""
        eng = self.local.FutEnv
        sides = []
        for tname, kcol, cols in ((ir["tables"][0], ir["col1"], ir["cols1"]), (ir["tables"][1], ir["col2"], ir["cols2"])):
            dev = self.local.tables[tname]._device
            need = [kcol] + [c for c in dict.fromkeys(cols) if c != kcol]
            n = dev.shape[0]
            ptrs, dts = [dev.device_ptr(c) for c in need], [dev.dtype(c) for c in need]
            if self.device_exchange:
                recv, nrecv = repartition_device(eng, ptrs, dts, n, 0, self.device, self.world)
                t = eng.table_from_device(nrecv, [c.data_ptr() for c in recv], dts, keepalive=recv)
            else:
                t = eng.table_from_device(n, ptrs, dts, keepalive=dev)
            sides.append((t, {c: i for i, c in enumerate(need)}))
        (t1, m1), (t2, m2) = sides
        res = eng.join(t1, t2, 0, 0, [m1[c] for c in ir["cols1"]], [m2[c] for c in ir["cols2"]])
        tens, dts = result_tensors(res, self.device)
        cols = gather_columns(tens, device=self.device, np_dtypes=dts)
        left_pos = {c: i for i, c in reversed(list(enumerate(ir["cols1"])))}
        right_pos = {c: len(ir["cols1"]) + i for i, c in reversed(list(enumerate(ir["cols2"])))}
        out = [cols[left_pos[c] if s == 0 else right_pos[c]] for s, c in ir["order"]]
        names = [f"{ir['tables'][s]}.{self.local.tables[ir['tables'][s]].get_schema()[c]}" for s, c in ir["order"]]
        if "limit" in ir:
            out = [c[: ir["limit"]] for c in out]
        return names, out
