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
        if device_type == "cuda":
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    return rank, local, world


class _DevicePtr:
    """Exposes a raw device address to torch without a copy."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (int(ptr), False), "version": 3, "strides": None}


def tensor_from_ptr(ptr, n, dtype, device):
    """Zero-copy torch view of `n` elements at device address `ptr`."""
    import torch
    typestr = {"float64": "<f8", "int64": "<i8", "float32": "<f4", "int32": "<i4", "uint32": "<u4"}[np.dtype(dtype).name]
    return torch.as_tensor(_DevicePtr(ptr, n, typestr), device=device)


def allreduce_partials(sum_t, cnt_t, group=None):
    """Merge per-shard dense GROUP BY partials in place: SUM over ranks of the
    f64 sums and of the i64 counts (the RCCL all-reduce of SURVEY.md 8(e))."""
    import torch.distributed as dist
    if dist.is_initialized():          # also with one rank: keeps the single-GPU run on the same code path
        dist.all_reduce(sum_t, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(cnt_t, op=dist.ReduceOp.SUM, group=group)
    return sum_t, cnt_t


def allreduce_minmax(min_t, max_t, group=None):
    """MIN / MAX partials (ncclMin / ncclMax)."""
    import torch.distributed as dist
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(min_t, op=dist.ReduceOp.MIN, group=group)
        dist.all_reduce(max_t, op=dist.ReduceOp.MAX, group=group)
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
    finish.  `eng`/`plan` are this rank's Engine and FgbPlan."""

    def __init__(self, eng, plan, device):
        self.eng, self.plan, self.device = eng, plan, device
        s_ptr, c_ptr = plan.acc_ptrs()
        self.sum_t = tensor_from_ptr(s_ptr, plan.G, np.float64, device)
        self.cnt_t = tensor_from_ptr(c_ptr, plan.G, np.int64, device)

    def step(self, p, cmp, thr, k, v, n, sum_out=None, count_out=None):
        self.plan.reset()
        self.plan.run(p, cmp, thr, k, v, n)
        allreduce_partials(self.sum_t, self.cnt_t)
        self.plan.finish(sum_out, count_out)
