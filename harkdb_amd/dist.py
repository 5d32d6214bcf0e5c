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
        if async_op and dist.get_backend(group) != "nccl":
            w1.wait()                                                 # gloo has no stream to order the two on: host-side wait
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


def allreduce_partials(sum_t, cnt_t, group=None, wait=True, how=None):
    """Merge per-shard dense GROUP BY partials in place: SUM over ranks of the
    f64 sums and of the i64 counts (the RCCL all-reduce of SURVEY.md 8(e)).  wait=False returns the pending
    work handles (call .wait() on each before reading the tensors): the collectives then run on RCCL's own
    stream beside whatever the caller enqueues next (ShardedFgb pipelines the next step's kernels under them).
    how: "allreduce" | "rs_ag" (see _allreduce_sum); None reads HARK_ALLREDUCE (default "allreduce") -- a caller that
    A/Bs the two passes `how` and leaves the environment alone."""
    import torch.distributed as dist
    works = []
    if dist.is_initialized():          # also with one rank: keeps the single-GPU run on the same code path
        if how is None:
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

    def __init__(self, eng, plan, device, plan2=None, acc_tensors=None, how=None):
        self.eng, self.plan, self.device = eng, plan, device
        self.how = how                                                    # form of the merge (allreduce_partials); None: HARK_ALLREDUCE
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
            allreduce_partials(self.sum_t, self.cnt_t, how=self.how)
            self.plan.finish(sum_out, count_out, check=True if check is None else bool(check))
            return
        i = self.turn
        self.turn = 1 - i
        plan, (sum_t, cnt_t) = self.plans[i], self.acc[i]
        plan.reset()
        plan.run(p, cmp, thr, k, v, n)
        works = allreduce_partials(sum_t, cnt_t, wait=False, how=self.how)         # runs beside the NEXT step's kernels
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
def gather_tensors(tens, group=None, device=None):
    """Concatenate per-rank 1-D torch tensors (one list entry per column, equal length on a rank) in rank order on every
    rank; returns (list of tensors on the collective's device, rows per rank).  One all-gather of the row counts, then ONE
    `all_gather_into_tensor` of a byte buffer that holds every column padded to the longest shard -- device tensors travel
    over RCCL as they are (no pickling, no host copy); under gloo (CPU tests, ranks sharing one GPU) they are staged
    through the host."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return list(tens), [int(tens[0].numel()) if tens else 0]
    world = dist.get_world_size(group)
    staged = dist.get_backend() == "gloo"
    home = tens[0].device if tens else torch.device("cpu")
    if not staged and device is None:
        device = torch.device("cuda", torch.cuda.current_device())        # RCCL moves device memory only
    dev = torch.device("cpu") if staged else torch.device(device)
    tens = [t.to(dev).contiguous() for t in tens]
    n_local = int(tens[0].numel()) if tens else 0
    mine = torch.tensor([n_local], dtype=torch.int64, device=dev)
    allc = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(allc, mine, group=group)
    counts = [int(x) for x in allc.tolist()]
    maxn = max(counts)
    if maxn == 0 or not tens:
        return [t[:0].to(home) for t in tens], counts
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
    out, off = [], 0
    for t, w in zip(tens, widths):
        pieces = [recv[r * maxn * row_bytes + off: r * maxn * row_bytes + off + counts[r] * w] for r in range(world)]
        out.append(torch.cat(pieces).view(t.dtype).to(home))
        off += maxn * w
    return out, counts


def gather_columns(cols, group=None, device=None, np_dtypes=None):
    """Concatenate per-rank result columns in rank order on every rank (row order = the unsharded table's order) and
    return them as numpy arrays: gather_tensors + one download.  `cols` may be numpy arrays (uploaded to `device` first
    when it is a GPU) or 1-D torch tensors.  HARK_GATHER=object selects the old `all_gather_object` path (A/B on
    hardware).  `np_dtypes` (one per column) names the dtypes of the returned arrays; without it they are derived from
    the local tensors, which cannot tell uint32 from int32 (torch carries u32 bit patterns in int32 tensors) -- a rank
    with an empty shard would otherwise answer with another dtype than its peers."""
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
        device = torch.device("cuda", torch.cuda.current_device())
    dev = torch.device("cpu") if staged else torch.device(device)
    dts, tens = [], []
    for c in cols:
        if isinstance(c, torch.Tensor):
            dts.append(np.dtype(str(c.dtype).replace("torch.", "")))
            tens.append(c.to(dev).contiguous())
        else:
            a = np.ascontiguousarray(c)
            dts.append(a.dtype)
            tens.append(torch.from_numpy(a.view(np.int32) if a.dtype == np.uint32 else a).to(dev))
    if np_dtypes is not None:
        dts = [np.dtype(d) for d in np_dtypes]
    out, _ = gather_tensors(tens, group, dev)
    return [t.cpu().numpy().view(dt) for t, dt in zip(out, dts)]


def result_tensors(res, device, limit=None):
    """The columns of a device-resident Result as zero-copy torch tensors (the first `limit` rows)."""
    n, m = res.shape
    if limit is not None:
        n = min(n, max(int(limit), 0))
    return [tensor_from_ptr(res.device_ptr(j), n, res.dtype(j), device) if n else __import__("torch").empty(0, dtype=getattr(__import__("torch"), _TORCH_OF[np.dtype(res.dtype(j)).name]), device=device)
            for j in range(m)], [np.dtype(res.dtype(j)) for j in range(m)]


def _reduce_tensor(t, op, group=None):
    """All-reduce of one tensor in place (RCCL on device tensors; under gloo device tensors are staged through the host)."""
    import torch.distributed as dist
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        x, back = _host_staged(t)
        dist.all_reduce(x, op={"sum": dist.ReduceOp.SUM, "min": dist.ReduceOp.MIN, "max": dist.ReduceOp.MAX}[op], group=group)
        back()
    return t


def merge_slot_columns(cols, kinds, np_dtypes, cnt, group=None):
    """Merge the per-key-slot partial aggregates of every rank elementwise (dense GROUP BY over shards).

    cols[j]: 1-D tensor of G slots holding this rank's partial of kind kinds[j] ("sum" | "min" | "max"), typed
    np_dtypes[j] (uint32 travels as its int32 bit pattern); cnt: int64 tensor, rows per slot on this rank.  Slots a rank
    has no row for hold unspecified values: they are replaced by the operator's neutral element first.  SUMs of f32 are
    widened to f64 for the reduction (the merged value does not depend on how rows were spread over the GPUs beyond one
    rounding of each shard's partial); unsigned MIN / MAX reduce as int64.  Returns [merged cols..., merged counts]."""
    import torch
    empty = cnt == 0
    out = []
    for t, kind, dt in zip(cols, kinds, np_dtypes):
        dt = np.dtype(dt)
        if kind == "sum":
            x = t.to(torch.float64) if dt.kind == "f" else t.to(torch.int64)
            x = torch.where(empty, torch.zeros_like(x), x)
            out.append(_reduce_tensor(x, "sum", group))
            continue
        if dt == np.dtype(np.uint32):
            x = t.to(torch.int64) & 0xFFFFFFFF
            neutral = 0xFFFFFFFF if kind == "min" else 0
        elif dt.kind == "f":
            x, neutral = t.clone(), float("inf") if kind == "min" else float("-inf")
        else:
            info = np.iinfo(dt)
            x, neutral = t.clone(), int(info.max if kind == "min" else info.min)
        x = torch.where(empty, torch.full_like(x, neutral), x)
        x = _reduce_tensor(x, kind, group)
        out.append(x.to(torch.int32) if dt == np.dtype(np.uint32) else x)        # back to the bit pattern (wraps above 2^31)
    total = _reduce_tensor(cnt.clone(), "sum", group)
    return out + [total]


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


def repartition_device(eng, ptrs, dtypes, n, key_index, device, world, group=None, splitters=None, descending=False, key_dtype=None):
    """Partition n rows held in device columns (raw pointers `ptrs`) by column
    `key_index` on the GPU and all-to-all them: by key hash
    (hark_op_partition_by_hash) or, when `splitters` is given, by key range
    (hark_op_partition_by_range: rank r receives the r-th range, rows of one
    source in table order).  `key_dtype` reads the key column's bits as another
    dtype of the same width (the join orders 32-bit keys as u32, join.fut:52).
    Returns (torch tensors owning the received columns, rows)."""
    import torch
    perm = torch.empty(max(n, 1), dtype=torch.int32, device=device)
    kdt = dtypes[key_index] if key_dtype is None else key_dtype
    if splitters is None:
        counts = eng.partition_by_hash(ptrs[key_index], kdt, n, world, perm.data_ptr())
    else:
        counts = eng.partition_by_range(ptrs[key_index], kdt, n, splitters, descending, perm.data_ptr())
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
    """All ranks pool their key samples (one tensor all-gather, no pickling) and derive the SAME splitters."""
    import torch
    import torch.distributed as dist
    local_sample = np.ascontiguousarray(local_sample)
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return choose_splitters(local_sample, world)
    dt = local_sample.dtype
    carrier = {4: np.int32, 8: np.int64}[dt.itemsize]                      # bit patterns travel; u32 / f32 keep their order on return
    pooled = gather_columns([local_sample.view(carrier)], group)[0].view(dt)
    return choose_splitters(pooled, world)


_SECOND_LEVEL = {"sum": "sum", "count": "sum", "min": "min", "max": "max", "prod": "prod"}


class TensorResult:
    """Device columns held by torch tensors behind the read-only interface of engine.Result (shape / dtype / device_ptr /
    columns): what the merged groups of all shards look like to the common HAVING / ORDER BY / LIMIT tail."""

    def __init__(self, tensors, np_dtypes, keep=None):
        self._t, self._dt, self._keep = [t.contiguous() for t in tensors], [np.dtype(d) for d in np_dtypes], keep

    @property
    def shape(self):
        return (int(self._t[0].numel()) if self._t else 0), len(self._t)

    def dtype(self, j):
        return self._dt[j]

    def device_ptr(self, j):
        return self._t[j].data_ptr() if self._t[j].numel() else 0

    def column(self, j, limit=None):
        n = self.shape[0]
        rows = n if limit is None else min(n, max(int(limit), 0))
        return self._t[j][:rows].cpu().numpy().view(self._dt[j])

    def columns(self, limit=None):
        return [self.column(j, limit) for j in range(len(self._t))]

    def free(self):
        self._t, self._keep = [], None


def shard_of_host_table(table_name, table, rank, world, local=False):
    """Rank `rank`'s rows of a host-side table source as a table.Table with the WHOLE table's schema (see
    ShardedFutharkContext.create_table).  No collective, no device: plain host work."""
    import pandas as pd
    from .table import Table, read_csv_byte_range
    if local:
        return Table(table_name, table)
    if isinstance(table, str) and table[-3:] == "csv":
        frame, headers = read_csv_byte_range(table, rank, world)
        return Table(table_name, frame)
    if isinstance(table, str):                                   # TXT: np.loadtxt has no row ranges; sliced after loading,
        full = Table(table_name, table)                          # under the WHOLE table's headers (c1, c2, ...: a bare ndarray
        table = pd.DataFrame(full.get_data(), columns=full.get_schema())   # slice would be renamed col1, col2, ... by load_np)
    lo, hi = shard_range(len(table), rank, world)
    return Table(table_name, table.iloc[lo:hi] if isinstance(table, pd.DataFrame) else table[lo:hi])


class ShardedFutharkContext:
    """FutharkContext whose tables are row-range shards, one rank per GPU.

    create_table() ingests only this rank's rows (a byte range of a CSV file, a slice of a
    DataFrame / ndarray, a per-rank source, or device columns); sql() runs the local operators
    on the shard and merges on the device: projection / WHERE results concatenate in rank
    order, GROUP BY partials are all-reduced per key slot (dense keys) or repartitioned to the
    owner of hash(key), ORDER BY is a sample sort, JOIN a range partition (reference order)."""

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

    def create_table(self, table_name, table, local=False):
        """The reference's call (FutharkContext.py:44-50) over shards.  `table` is a file name, a DataFrame or an ndarray.

        local=False: `table` names the WHOLE table on every rank and each rank ingests ITS rows only -- a CSV file is cut
        into world byte ranges at row boundaries and a rank parses just its range (table.read_csv_byte_range: nobody
        reads or parses the whole file); a DataFrame / ndarray / TXT file is sliced by shard_range before any conversion.
        local=True: `table` holds this rank's rows only (e.g. one file per rank); the table is the rank-order
        concatenation.  Either way the ranks agree on every column's dtype (the widest any shard needs) with one MAX
        all-reduce and on the global row ranges with one all-gather of the row counts."""
        import pandas as pd
        from .table import dtype_code, DTYPE_CODES
        part = shard_of_host_table(table_name, table, self.rank, self.world, local)
        codes = [dtype_code(d) for d in part.column_dtypes()] if part.get_data().shape[0] else [0] * len(part.get_schema())
        codes = self._allreduce_small(codes, "max")
        cols = part.host_columns([DTYPE_CODES[c] for c in codes])
        self.local.create_table(table_name, pd.DataFrame({h: c for h, c in zip(part.get_schema(), cols)}))
        self._register_rows(table_name, len(cols[0]) if cols else 0)

    def create_table_from_device(self, table_name, schema, ptrs, dtypes, n_local, keepalive=None):
        """This rank's rows as columns that already live in ITS GPU's memory (raw device addresses, 16-byte aligned; e.g.
        generated there, or loaded by another library): no host copy anywhere.  Rank r's rows follow rank r - 1's."""
        self.local.create_table_from_device(table_name, schema, ptrs, dtypes, int(n_local), keepalive=keepalive)
        self._register_rows(table_name, int(n_local))

    def _register_rows(self, table_name, n_local):
        import torch
        import torch.distributed as dist
        counts = [n_local]
        if dist.is_initialized() and dist.get_world_size() > 1:
            dev = "cpu" if dist.get_backend() == "gloo" else self.device
            mine = torch.tensor([n_local], dtype=torch.int64, device=dev)
            allc = torch.empty(self.world, dtype=torch.int64, device=dev)
            dist.all_gather_into_tensor(allc, mine)
            counts = [int(x) for x in allc.tolist()]
        lo = sum(counts[: self.rank])
        self.rows[table_name] = (sum(counts), lo, lo + n_local)

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
        return self._run(sql_parse(self.local.tables, sql_statement))

    def _run(self, ir):
        """Executes a planned statement (the IR of parse.sql_parse_tree) over the shards."""
        if ir.get("join"):
            return self._join(ir)
        if "groupbys" not in ir:
            if ("orderby" in ir or ir.get("orderby_all")) and (self.world > 1 or self.device_exchange):
                return self._orderby(ir)
            limit = ir.pop("limit", None)
            names, res = self.local.select_result_ir(ir)                # device-resident: gathered over RCCL as it is
            tens, dts = result_tensors(res, self.device, limit)         # no rank contributes more than LIMIT rows
            cols = gather_columns(tens, device=self.device, np_dtypes=dts)
            return names, ([c[:limit] for c in cols] if limit is not None else cols)
        return self._groupby(ir)

    def _allreduce_small(self, values, op):
        """Elementwise MIN / MAX / SUM over ranks of a short list of Python ints (statistics, not data): one int64 tensor."""
        import torch
        import torch.distributed as dist
        if not dist.is_initialized() or dist.get_world_size() == 1:
            return [int(v) for v in values]
        dev = "cpu" if dist.get_backend() == "gloo" else self.device
        t = torch.tensor([int(v) for v in values], dtype=torch.int64, device=dev)
        dist.all_reduce(t, op={"min": dist.ReduceOp.MIN, "max": dist.ReduceOp.MAX, "sum": dist.ReduceOp.SUM}[op])
        return [int(x) for x in t.tolist()]

    def _global_key_ranges(self, dev, g_cols):
        """(mins, spans) of 32-bit integer key columns over ALL shards (one MIN and one MAX all-reduce of 2 x keys int64
        words; an empty shard contributes the neutral elements), or None when a key column is not a 32-bit integer."""
        if any(np.dtype(dev.dtype(c)) not in (np.dtype(np.int32), np.dtype(np.uint32)) for c in g_cols):
            return None
        eng, big = self.local.FutEnv, 1 << 62
        mine = [eng.column_range(dev, c) for c in g_cols] if dev.shape[0] > 0 else [(big, -big)] * len(g_cols)
        mins = self._allreduce_small([lo for lo, _ in mine], "min")
        maxs = self._allreduce_small([hi for _, hi in mine], "max")
        if any(lo > hi for lo, hi in zip(mins, maxs)):                   # every shard is empty
            return [0] * len(g_cols), [1] * len(g_cols)
        return mins, [hi - lo + 1 for lo, hi in zip(mins, maxs)]

    def _groupby(self, ir):
        """GROUP BY over row-range shards.  The statement is planned exactly as on one GPU (FutharkContext.
        _groupby_extended: aggregates, HAVING, ORDER BY, LIMIT, composite keys encoded with the key ranges of ALL shards);
        only the aggregation itself is replaced by a provider that merges the shards' partial aggregates ON THE DEVICE:

          dense key domain (<= 2^21 slots)   every rank lays its partials out per key slot (hark_entry_filter_groupby_slots)
                                             and the slots are merged elementwise with all-reduces (SUM / MIN / MAX);
          anything else                      partial aggregates are repartitioned by hash(key) with an all-to-all, folded by
                                             their owner, and the owners' groups all-gathered as device tensors.

        Either way every rank then holds the merged groups in HBM, and HAVING / ORDER BY / LIMIT run there
        (hark_entry_topk, or compaction + radix sort); only the final rows are downloaded."""
        tab = self.local.tables[ir["table_name"]]
        dev, schema = tab._device, tab.get_schema()
        g_cols = ir.get("g_cols", [ir["g_col"]])
        ranges = self._global_key_ranges(dev, g_cols)
        n_global = self.rows.get(ir["table_name"], (dev.shape[0],))[0]
        self.last_groupby_path = None

        def provider(cur, dev_preds, gkey, specs):
            res = None
            if ranges is not None and not os.environ.get("HARK_NO_DENSE_MERGE"):
                res = self._merged_dense(cur, dev_preds, gkey, specs, ranges, len(g_cols) > 1, n_global)
            if res is None:
                res = self._merged_by_owner(cur, dev_preds, gkey, specs)
            return res

        return self.local._groupby_extended(dev, schema, ir, provider=provider, key_ranges=ranges if len(g_cols) > 1 else None,
                                            subset_provider=self._subset_merged)

    def _subset_merged(self, cur, dev_preds, gkey, keys, specs):
        """Late aggregation over shards: the aggregates `specs` of the groups `keys` (the LIMIT survivors of the first phase,
        the same on every rank) -- every rank aggregates its own rows of those groups (hark_entry_filter_groupby_subset),
        the len(keys)-row partials are merged by all-reduce (SUM / MIN / MAX; AVG = SUM and COUNT).  numpy columns."""
        import torch
        eng = self.local.FutEnv
        self.late_aggregations = getattr(self, "late_aggregations", 0) + 1
        part = []
        for f, c in specs:
            ps = ("sum" if f == "avg" else f, c)
            if f != "count" and ps not in part:
                part.append(ps)
        r = eng.filter_groupby_subset(cur, dev_preds, gkey, keys, part + [("count", 0)])
        cols = r.columns()
        dts = [np.dtype(c.dtype) for c in cols[:-1]]
        tens = [torch.from_numpy(np.ascontiguousarray(c).view(np.int32) if c.dtype == np.uint32 else np.ascontiguousarray(c)).to(self.device) for c in cols[:-1]]
        cnt = torch.from_numpy(np.ascontiguousarray(cols[-1]).astype(np.int64)).to(self.device)
        merged = merge_slot_columns(tens, [f for f, _ in part], dts, cnt)
        total = merged[-1]
        out = []
        for f, c in specs:
            if f == "count":
                out.append(total.cpu().numpy())
                continue
            j = part.index(("sum" if f == "avg" else f, c))
            t, dt = merged[j], dts[j]
            if f == "avg":                                              # f32(sum / count), as the single-GPU read-out computes it
                out.append((t.to(torch.float64) / total.to(torch.float64).clamp(min=1.0)).to(torch.float32).cpu().numpy())
            elif f == "sum" and dt.kind == "f":
                out.append(t.to(torch.float32).cpu().numpy())          # merged in f64, rounded once
            else:
                a = t.cpu().numpy()
                out.append(a.view(np.uint32) if dt == np.dtype(np.uint32) and a.dtype == np.int32 else a.astype(dt) if f != "sum" else a)
        return out

    # ---- dense key domain: all-reduce of per-slot partial aggregates ---------------
    def _merged_dense(self, cur, dev_preds, gkey, specs, ranges, composite, n_global):
        """[key, aggregates...] of all groups, merged over the shards by all-reduce; None when the shape is not dense."""
        import torch
        eng = self.local.FutEnv
        kdt = np.dtype(cur.dtype(gkey))
        if composite:                                                   # the composite column already lies in [0, prod(spans))
            if kdt != np.dtype(np.int32):
                return None
            base, G = 0, int(np.prod([int(x) for x in ranges[1]], dtype=object))
        else:
            base, G = int(ranges[0][0]), int(ranges[0][0]) + int(ranges[1][0])     # slots 0 .. max key
            if base < 0:
                return None                                             # negative keys: not a slot index (owner path)
        if G > (1 << 21) or G > 8 * max(int(n_global), 1) + 4096:
            return None
        four = (np.dtype(np.float32), np.dtype(np.int32), np.dtype(np.uint32))
        if any(f not in ("sum", "avg", "min", "max", "count") or (f != "count" and np.dtype(cur.dtype(c)) not in four) for f, c in specs):
            return None
        # what travels: SUM / MIN / MAX per (operator, column); COUNT comes with every pass; AVG = SUM and COUNT
        # (specs name COUNT(*) as ("count", 0): the column of a count is never read)
        part = []
        for f, c in specs:
            ps = ("sum" if f == "avg" else f, c)
            if f != "count" and ps not in part:
                part.append(ps)
        res = eng.filter_groupby_slots(cur, dev_preds, gkey, G, part)
        if res is None:
            return None
        tens, dts = result_tensors(res, self.device)
        merged = merge_slot_columns(tens[:-1], [f for f, _ in part], dts[:-1], tens[-1])
        cnt = merged[-1]
        cols, np_dts = [torch.arange(G, dtype=torch.int32, device=cnt.device)], [kdt]
        for f, c in specs:
            if f == "count":
                cols.append(cnt); np_dts.append(np.dtype(np.int64))
                continue
            j = part.index(("sum" if f == "avg" else f, c))
            t, dt = merged[j], dts[j]
            if f == "avg":                                              # f32(sum / count), as the fused read-out computes it
                t, dt = (t.to(torch.float64) / cnt.to(torch.float64).clamp(min=1.0)).to(torch.float32), np.dtype(np.float32)
            elif t.dtype == torch.float64:
                t = t.to(torch.float32)                                 # SUM of an f32 column: merged in f64, rounded once
            cols.append(t.contiguous()); np_dts.append(dt)
        cols.append(cnt); np_dts.append(np.dtype(np.int64))
        t = eng.table_from_device(G, [c.data_ptr() for c in cols], np_dts, keepalive=(cols, res))
        out = eng.filter_sel(t, [(len(cols) - 1, ">", 0)], cols=list(range(len(cols) - 1)), want_row_index=False)   # the non-empty groups, ascending key
        out._keep = (t,)
        self.last_groupby_path = "dense all-reduce"
        return out

    # ---- any keys: all-to-all of partial aggregates to the owner of hash(key) ---------
    def _merged_by_owner(self, cur, dev_preds, gkey, specs):
        import torch
        eng = self.local.FutEnv
        specs = [(f, 0 if f == "count" else c) for f, c in specs]       # COUNT(*): the column is never read
        part = []
        for f, c in specs:
            for ps in ((("sum", c), ("count", 0)) if f == "avg" else ((f, c),)):
                if ps not in part:
                    part.append(ps)
        res = eng.filter_groupby(cur, dev_preds, gkey, part)            # this shard's groups: [key, partials...]
        n, m = res.shape
        dts = [np.dtype(res.dtype(j)) for j in range(m)]
        if self.world > 1 or self.device_exchange:
            recv, nrecv = repartition_device(eng, [res.device_ptr(j) for j in range(m)], dts, n, 0, self.device, self.world)
            t = eng.table_from_device(nrecv, [c.data_ptr() for c in recv], dts, keepalive=(recv, res))
            own = eng.filter_groupby(t, None, 0, [(_SECOND_LEVEL[f], 1 + j) for j, (f, _) in enumerate(part)])     # complete groups of the keys this rank owns
            own._keep = (t,)
            tens, dts = result_tensors(own, self.device)
            allt, _ = gather_tensors(tens, device=self.device)          # every rank: all groups, owner by owner (device tensors)
            g = int(allt[0].numel())
            t2 = eng.table_from_device(g, [x.data_ptr() for x in allt], dts, keepalive=(allt, own))
            res = eng.sort(t2, 0, list(range(len(allt))))               # ascending key (the key column's own order, as on one GPU)
            res._keep = (t2,)
            n, m = res.shape
            dts = [np.dtype(res.dtype(j)) for j in range(m)]
        # [key, partials...] -> [key, aggregates in spec order]
        tens, _ = result_tensors(res, self.device)
        cols, np_dts = [tens[0]], [dts[0]]
        for f, c in specs:
            if f == "avg":
                sj, cj = 1 + part.index(("sum", c)), 1 + part.index(("count", 0))
                cols.append((tens[sj].to(torch.float64) / tens[cj].to(torch.float64).clamp(min=1.0)).to(torch.float32)); np_dts.append(np.dtype(np.float32))
            else:
                j = 1 + part.index((f, c))
                cols.append(tens[j]); np_dts.append(dts[j])
        self.last_groupby_path = "owner all-to-all"
        return TensorResult(cols, np_dts, keep=res)

    # ---- RCCL all-to-all paths ------------------------------------------------------
    def _sample_keys(self, ptr, dt, n):
        """Up to SAMPLES_PER_RANK evenly spaced values of a device column, on the host."""
        import torch
        pos = sample_positions(n)
        if not pos.size:
            return np.empty(0, dtype=np.dtype(dt))
        idx = torch.as_tensor(pos.astype(np.int32), device=self.device)
        buf = torch.empty(pos.size, dtype=getattr(torch, _TORCH_OF[np.dtype(dt).name]), device=self.device)
        self.local.FutEnv.gather(ptr, dt, idx.data_ptr(), buf.data_ptr(), pos.size)
        return buf.cpu().numpy().view(np.dtype(dt))

    def _orderby(self, ir):
        """SELECT ... [WHERE ...] ORDER BY col [, col ...] [DESC] [LIMIT n] over row-range shards as a
        sample sort: local WHERE, pooled key samples -> splitters, range partition on the
        GPU, all-to-all, local stable radix sort, ranges concatenated in rank order.  Ties
        keep table order: a range receives its rows by (source rank, local position).
        Several sort keys are folded into ONE composite key column first (hark_table_composite_key with the key ranges
        of ALL shards, so that every rank encodes alike; ascending composite = lexicographic order of the tuple)."""
        eng, loc = self.local.FutEnv, self.local
        tab = loc.tables[ir["table_name"]]
        dev, schema = tab._device, tab.get_schema()
        many = ir.get("orderby_all")
        if many:
            if any(sp[0] != "col" for sp, _ in many) or len({d for _, d in many}) != 1:
                raise Exception("ORDER BY on several keys takes plain columns, all ascending or all descending")
            okeys, desc = [sp[1] for sp, _ in many], bool(many[0][1])
        else:
            okeys, desc = [ir["orderby"][0][1]], bool(ir["orderby"][1])
        need = list(dict.fromkeys(okeys + list(ir["select"])))
        if ir.get("where"):
            cur, cmap = loc._filtered(dev, ir["where"], set(need))
        else:
            cur, cmap = dev, {c: c for c in need}
        n = cur.shape[0]
        ptrs, dts = [cur.device_ptr(cmap[c]) for c in need], [cur.dtype(cmap[c]) for c in need]
        keep = [cur]
        if len(okeys) > 1:
            ranges = self._global_key_ranges(dev, okeys)                 # of the unfiltered shards: a superset, the same on every rank
            if ranges is None:
                raise Exception("ORDER BY on several keys takes 32-bit integer columns")
            buf, cdt, _, _ = eng.composite_key(cur, [cmap[c] for c in okeys], ranges=ranges)
            keep.append(buf)
            ptrs, dts = [buf.ptr or 0] + ptrs, [cdt] + dts              # the sort key travels in front of the needed columns
            sel = [1 + need.index(c) for c in ir["select"]]
        else:
            sel = [need.index(c) for c in ir["select"]]
        splitters = gather_splitters(self._sample_keys(ptrs[0], dts[0], n), self.world)
        recv, nrecv = repartition_device(eng, ptrs, dts, n, 0, self.device, self.world, splitters=splitters, descending=desc)
        t = eng.table_from_device(nrecv, [c.data_ptr() for c in recv], dts, keepalive=(recv, keep))
        res = eng.sort(t, 0, sel, descending=desc)
        tens, dts = result_tensors(res, self.device, ir.get("limit"))
        cols = gather_columns(tens, device=self.device, np_dtypes=dts)
        if "limit" in ir:
            cols = [c[: ir["limit"]] for c in cols]
        return [schema[c] for c in ir["select"]], cols

    def _join(self, ir):
        """Two-table FROM over shards, in the REFERENCE's row order (futhark/join.fut:52-75: ascending key -- 32-bit keys
        as u32 bit patterns --, then left row, then right row).  Both sides are RANGE-partitioned by the join key with
        the same world - 1 splitters (quantiles of key samples pooled from both sides) and exchanged with an all-to-all;
        rank r owns the r-th key range, so the owners' results concatenated in rank order ascend by key, and equal keys
        never straddle two owners (part = number of splitters <= key).  Inside an owner the received rows keep (source
        rank, local position) = the unsharded tables' row order, so the local join's (key, left row, right row) order
        is the global one."""
        eng = self.local.FutEnv
        exchange = self.world > 1 or self.device_exchange
        sides, samples = [], []
        for tname, kcol, cols, where in ((ir["tables"][0], ir["col1"], ir["cols1"], ir.get("where1")), (ir["tables"][1], ir["col2"], ir["cols2"], ir.get("where2"))):
            dev = self.local.tables[tname]._device
            need = [kcol] + [c for c in dict.fromkeys(cols) if c != kcol]
            if where:                                                    # conjuncts of the WHERE on this table: below the exchange and the join
                cur, cmap = self.local._filtered(dev, where, set(need))
                dev, need_at = cur, [cmap[c] for c in need]
            else:
                need_at = need
            n = dev.shape[0]
            ptrs, dts = [dev.device_ptr(c) for c in need_at], [dev.dtype(c) for c in need_at]
            kdt = np.dtype(np.uint32) if np.dtype(dts[0]).itemsize == 4 else np.dtype(dts[0])      # the join's own key order
            sides.append((dev, need, n, ptrs, dts, kdt))
            if exchange:
                samples.append(self._sample_keys(ptrs[0], dts[0], n).view(kdt))
        if exchange and samples[0].dtype != samples[1].dtype:
            raise Exception("join keys must have the same width on both sides")
        splitters = gather_splitters(np.concatenate(samples), self.world) if exchange else None
        tabs = []
        for dev, need, n, ptrs, dts, kdt in sides:
            if exchange:
                recv, nrecv = repartition_device(eng, ptrs, dts, n, 0, self.device, self.world, splitters=splitters, key_dtype=kdt)
                t = eng.table_from_device(nrecv, [c.data_ptr() for c in recv], dts, keepalive=recv)
            else:
                t = eng.table_from_device(n, ptrs, dts, keepalive=dev)
            tabs.append((t, {c: i for i, c in enumerate(need)}))
        (t1, m1), (t2, m2) = tabs
        res = eng.join(t1, t2, 0, 0, [m1[c] for c in ir["cols1"]], [m2[c] for c in ir["cols2"]])
        if "post" in ir:
            # the clauses around the join run over its result as over any sharded table: every owner's pairs are its rows
            # (key ranges ascend with the rank, so rank order IS the join's order)
            from .parse import sql_parse_tree, JOIN_RESULT
            n, m = res.shape
            self.create_table_from_device(JOIN_RESULT, ir["post_schema"], [res.device_ptr(j) for j in range(m)], [res.dtype(j) for j in range(m)], n,
                                          keepalive=(res, tabs))
            try:
                return self._run(sql_parse_tree(self.local.tables, ir["post"]))
            finally:
                self.drop_table(JOIN_RESULT)
        tens, dts = result_tensors(res, self.device, ir.get("limit"))     # no owner contributes more than LIMIT rows
        cols = gather_columns(tens, device=self.device, np_dtypes=dts)
        left_pos = {c: i for i, c in reversed(list(enumerate(ir["cols1"])))}
        right_pos = {c: len(ir["cols1"]) + i for i, c in reversed(list(enumerate(ir["cols2"])))}
        out = [cols[left_pos[c] if s == 0 else right_pos[c]] for s, c in ir["order"]]
        names = [f"{ir['tables'][s]}.{self.local.tables[ir['tables'][s]].get_schema()[c]}" for s, c in ir["order"]]
        if "limit" in ir:
            out = [c[: ir["limit"]] for c in out]
        return names, out
