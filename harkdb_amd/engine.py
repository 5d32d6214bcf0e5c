"""Thin object layer over the C ABI: Engine (context), DeviceTable, Result.

Plays the role of `futhark_ffi.Futhark` in the reference
(FutharkContext.py:41, :65-66, :70-71): numpy in, entry call, numpy out --
except that tables are uploaded once and stay on the GPU.
"""
import ctypes as C

import numpy as np

from . import _ffi


_CANON = {"==": "=", "<>": "!="}


def normalise_predicate(dt, cmp, value):
    """`column <cmp> value` for a column of dtype `dt`, rewritten so that the constant is exactly representable in `dt`.

    The C entries read the constant AS the column's dtype (hark.h: hark_entry_filter_sel); a plain cast would truncate a
    fractional literal on an integer column (`x < 2.5` must keep x = 2) and wrap one outside the dtype's range
    (`x < 3000000000` on i32 must keep every row).  Returns (cmp, 1-element array of dtype dt).  Always-true becomes
    `>= dtype.min`, always-false `< dtype.min`."""
    import math
    dt = np.dtype(dt)
    cmp = _CANON.get(cmp, cmp)
    if cmp not in (">", ">=", "<", "<=", "=", "!="):
        raise KeyError(cmp)
    if dt.kind == "f":
        return cmp, np.asarray([value]).astype(dt)
    info = np.iinfo(dt)
    true, false = (">=", np.asarray([info.min], dtype=dt)), ("<", np.asarray([info.min], dtype=dt))
    if isinstance(value, (np.generic, np.ndarray)):
        value = np.asarray(value).reshape(-1)[0].item()
    if isinstance(value, bool):
        value = int(value)
    if isinstance(value, float):
        if math.isnan(value):
            return true if cmp == "!=" else false
        if math.isinf(value):
            below = value > 0                      # every column value lies below +inf
            return true if cmp == "!=" or (below and cmp in ("<", "<=")) or (not below and cmp in (">", ">=")) else false
        if value != math.floor(value):
            if cmp == "=":
                return false
            if cmp == "!=":
                return true
            cmp, value = ("<=", math.floor(value)) if cmp in ("<", "<=") else (">=", math.ceil(value))
        value = int(value)
    value = int(value)
    if value > info.max:
        return true if cmp in ("<", "<=", "!=") else false
    if value < info.min:
        return true if cmp in (">", ">=", "!=") else false
    return cmp, np.asarray([value], dtype=dt)


class PinnedBlock:
    """A pinned host block handed out by the library (hark_host_alloc and the *_pinned downloads): the storage of numpy
    arrays that the copy engine filled directly.  Exposes the array interface, so `np.asarray(block)` is a byte view whose
    base is this object -- every view / slice of it keeps the block alive, and the block goes back to the context's cache
    of pinned blocks when the last of them is collected."""

    def __init__(self, eng, ptr, nbytes):
        self._eng, self._ptr, self.nbytes = eng, int(ptr), int(nbytes)
        self.__array_interface__ = {"shape": (self.nbytes,), "typestr": "|u1", "data": (self._ptr, False), "version": 3}

    def array(self, offset, count, dtype):
        dtype = np.dtype(dtype)
        return np.asarray(self)[offset: offset + count * dtype.itemsize].view(dtype)

    def __del__(self):
        try:
            if self._ptr:
                # a block that outlives its context is released without it (include/hark.h: hark_host_free(NULL, p))
                self._eng.lib.hark_host_free(self._eng.ctx if self._eng.ctx is not None else None, self._ptr)
                self._ptr = 0
        except Exception:
            pass


PINNED_FROM = 1 << 16                     # downloads of at least this many bytes land in a pinned block (below: the context's pinned scratch)
PINNED_UPTO = 2 << 30                     # ... and of at most this many (pinned memory is a finite resource of the host)


class Result:
    """Device-resident query result (futhark opaque array + from_futhark)."""

    def __init__(self, eng, handle):
        self._eng, self._h = eng, handle

    @property
    def shape(self):
        n, m = C.c_int64(), C.c_int64()
        self._eng._chk(self._eng.lib.hark_result_shape(self._h, C.byref(n), C.byref(m)))
        return n.value, m.value

    def dtype(self, j):
        return _ffi.NP_OF[self._eng.lib.hark_result_dtype(self._h, j)]

    def column(self, j, limit=None):
        """Column j on the host; limit = n downloads only the first n rows (LIMIT is applied before the PCIe copy)."""
        n, _ = self.shape
        rows = n if limit is None else min(n, max(int(limit), 0))
        if rows == 0:
            return np.empty(0, dtype=self.dtype(j))
        return self._eng.download(self.device_ptr(j), rows, self.dtype(j))

    def columns(self, limit=None):
        """Every column on the host (the first `limit` rows).  Small results come through the context's pinned scratch with one
        synchronisation; larger ones are written by the copy engine straight into ONE pinned block that the returned arrays
        are views of (copies enqueued back to back, one synchronisation, no bounce buffer and no host memcpy)."""
        n, m = self.shape
        rows = n if limit is None else min(n, max(int(limit), 0))
        if m > 1 and rows > 0 and rows * 16 * m <= 65536:              # a small result or a LIMIT prefix: every column with one synchronisation
            outs = [np.empty(rows, dtype=self.dtype(j)) for j in range(m)]
            ptrs = (C.c_void_p * m)(*[o.ctypes.data for o in outs])
            self._eng._chk(self._eng.lib.hark_result_columns_prefix(self._eng.ctx, self._h, rows, ptrs))
            return outs
        dts = [np.dtype(self.dtype(j)) for j in range(m)]
        total = sum(rows * d.itemsize + 64 for d in dts)
        if m > 0 and rows > 0 and PINNED_FROM <= total <= PINNED_UPTO:
            blk, offs = C.c_void_p(), (C.c_int64 * m)()
            self._eng._chk(self._eng.lib.hark_result_columns_pinned(self._eng.ctx, self._h, rows, C.byref(blk), offs))
            block = PinnedBlock(self._eng, blk.value, max(offs[j] + rows * dts[j].itemsize for j in range(m)))
            return [block.array(offs[j], rows, dts[j]) for j in range(m)]
        return [self.column(j, limit) for j in range(m)]

    def matrix(self, cols=None, limit=None, dtype=None):
        """The reference's result shape (from_futhark, FutharkContext.py:66,71): ONE row-major [rows][len(cols)] matrix of the
        result columns `cols` (default: all; repeats allowed) whose element type is numpy's result_type of their dtypes,
        built ON THE DEVICE and copied once into a pinned block (hark_result_matrix_pinned)."""
        n, m = self.shape
        cols = list(range(m)) if cols is None else [int(c) for c in cols]
        for c in cols:
            if not 0 <= c < m:
                raise _ffi.HarkError(_ffi.EBOUNDS, f"result_matrix: column {c} of {m}")
        rows = n if limit is None else min(n, max(int(limit), 0))
        if dtype is None:
            dts = [np.dtype(self.dtype(j)) for j in cols]
            dtype = np.dtype(np.int32) if not dts else (dts[0] if len(set(dts)) == 1 else np.result_type(*dts))
        dtype = np.dtype(dtype)
        code = {np.dtype(np.int32): _ffi.I32, np.dtype(np.uint32): _ffi.U32, np.dtype(np.float32): _ffi.F32,
                np.dtype(np.int64): _ffi.I64, np.dtype(np.float64): _ffi.F64}[dtype]
        if rows == 0 or not cols:
            return np.empty((rows, len(cols)), dtype=dtype)
        if rows * len(cols) * dtype.itemsize <= PINNED_UPTO:
            blk = C.c_void_p()
            arr = (C.c_int32 * len(cols))(*cols)
            rc = self._eng.lib.hark_result_matrix_pinned(self._eng.ctx, self._h, arr, len(cols), rows, code, C.byref(blk))
            if rc == 0:
                block = PinnedBlock(self._eng, blk.value, rows * len(cols) * dtype.itemsize)
                return block.array(0, rows * len(cols), dtype).reshape(rows, len(cols))
            if rc not in (_ffi.ENOMEM, _ffi.EUNSUPPORTED):
                self._eng._chk(rc)
        # a matrix too large to pin (or to double in device scratch): typed columns through pageable memory, interleaved here
        out = np.empty((rows, len(cols)), dtype=dtype)
        have = {}
        for j, c in enumerate(cols):
            if c not in have:
                have[c] = self.column(c, rows)
            out[:, j] = have[c].view(dtype) if have[c].dtype.itemsize == dtype.itemsize and dtype.kind in "iu" and have[c].dtype.kind in "iu" else have[c]
        return out

    def device_ptr(self, j):
        return self._eng.lib.hark_result_column_device(self._h, j)

    def to_numpy(self, dtype=None):
        """Row-major [n][m] matrix like `from_futhark` (FutharkContext.py:66)."""
        n, m = self.shape
        if dtype is None:
            dts = {np.dtype(self.dtype(j)) for j in range(m)} or {np.dtype(np.int32)}
            dtype = dts.pop() if len(dts) == 1 else (np.float32 if np.dtype(np.float32) in dts else np.int64)
        dtype = np.dtype(dtype)
        out = np.empty((n, m), dtype=dtype)
        if n and m:
            self._eng._chk(self._eng.lib.hark_result_values_2d(self._eng.ctx, self._h, out.ctypes.data, _ffi.DT_OF[dtype]))
        return out

    def free(self):
        if self._h is not None:
            self._eng.lib.hark_result_free(self._eng.ctx, self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceTable:
    def __init__(self, eng, handle, keepalive=None):
        self._eng, self._h, self._keep = eng, handle, keepalive

    @property
    def shape(self):
        n, m = C.c_int64(), C.c_int64()
        self._eng._chk(self._eng.lib.hark_table_shape(self._h, C.byref(n), C.byref(m)))
        return n.value, m.value

    def dtype(self, j):
        return _ffi.NP_OF[self._eng.lib.hark_table_dtype(self._h, j)]

    def device_ptr(self, j):
        return self._eng.lib.hark_table_column_device(self._h, j)

    def invalidate_stats(self, col=-1):
        """Forget the statistics cached with column `col` (default: every column).  Tables are immutable; a caller that
        rewrites a column it lent with table_from_device calls this before the next query (include/hark.h)."""
        self._eng._chk(self._eng.lib.hark_table_invalidate_stats(self._eng.ctx, self._h, int(col)))

    def free(self):
        if self._h is not None:
            self._eng.lib.hark_table_free(self._eng.ctx, self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceBuffer:
    """A pool block returned by the library (freed with hark_dev_free when dropped)."""

    def __init__(self, eng, ptr):
        self._eng, self.ptr = eng, ptr

    def free(self):
        if self.ptr:
            self._eng.free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            if self._eng.ctx is not None:
                self.free()
        except Exception:
            pass


class FgbPlan:
    """Workspace + knobs of the fused filter->group-by (hark_fgb_plan)."""

    def __init__(self, eng, max_rows, G, **knobs):
        self._eng, self.G = eng, int(G)
        h = C.c_void_p()
        eng._chk(eng.lib.hark_fgb_plan_new(eng.ctx, C.byref(h), int(max_rows), int(G)))
        self._h = h
        for k, v in knobs.items():
            self.set(k, v)

    def set(self, key, value):
        rc = self._eng.lib.hark_fgb_plan_set(self._h, key.encode(), int(value))
        if rc:
            raise _ffi.HarkError(rc, f"fgb_plan_set({key}={value}) rejected")

    def reset(self):
        self._eng._chk(self._eng.lib.hark_fgb_reset(self._eng.ctx, self._h))

    MAX_ROWS_PER_CALL = 1 << 31      # the C entry takes < 2^32 rows; larger shards are fed in 16-byte-aligned pieces

    def run(self, p, cmp, thr, k, v, n):
        """Accumulate one batch.  All pointers are raw device addresses (ints); p = None: no filter,
        v = None: COUNT only (no value column is read)."""
        op = _ffi.CMP[cmp] if isinstance(cmp, str) else int(cmp)
        pstep = 1 / 8 if op == _ffi.CMP["mask"] else 4         # cmp "mask": p is a survivor bitmask, one bit per row
        for lo in range(0, int(n), self.MAX_ROWS_PER_CALL):
            m = min(self.MAX_ROWS_PER_CALL, int(n) - lo)
            self._eng._chk(self._eng.lib.hark_op_filter_groupby_dense_f32(
                self._eng.ctx, self._h, None if p is None else p + int(pstep * lo), op, float(thr), k + 4 * lo, None if v is None else v + 4 * lo, m))

    def acc_ptrs(self):
        """(device address of double[G] sums, device address of int64[G] counts)."""
        s, c = C.c_void_p(), C.c_void_p()
        self._eng._chk(self._eng.lib.hark_fgb_acc_device(self._h, C.byref(s), C.byref(c)))
        return s.value, c.value

    def finish(self, sum_ptr=None, count_ptr=None, check=True):
        """Read the accumulators out; check=False leaves the sticky error word on the device (no host round trip):
        call check() once after the last step."""
        fn = self._eng.lib.hark_fgb_finish if check else self._eng.lib.hark_fgb_finish_async
        self._eng._chk(fn(self._eng.ctx, self._h, sum_ptr, count_ptr))

    def check(self):
        self._eng._chk(self._eng.lib.hark_fgb_check(self._eng.ctx, self._h))

    def timing(self):
        """({kind: ms}, {kind: launches}) since the last call; needs set('timing', 1)."""
        ms, cnt = (C.c_double * 3)(), (C.c_int64 * 3)()
        self._eng._chk(self._eng.lib.hark_fgb_timing(self._eng.ctx, self._h, ms, cnt))
        kinds = ("single", "producer", "consumer")
        return {k: ms[i] for i, k in enumerate(kinds)}, {k: cnt[i] for i, k in enumerate(kinds)}

    def free(self):
        if self._h is not None:
            self._eng.lib.hark_fgb_plan_free(self._eng.ctx, self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Engine:
    def __init__(self, device=0):
        self.lib = _ffi.load()
        ctx = C.c_void_p()
        rc = self.lib.hark_context_new(C.byref(ctx), int(device))
        if rc:
            raise _ffi.HarkError(rc, f"hark_context_new(device={device}) failed: no usable MI355X/HIP device "
                                     "(libhark has no CPU fallback)")
        self.ctx = ctx
        self.device = int(device)

    # -- plumbing --------------------------------------------------------------
    def _chk(self, rc):
        if rc:
            raise _ffi.HarkError(rc, (self.lib.hark_context_get_error(self.ctx) or b"").decode())

    def sync(self):
        self._chk(self.lib.hark_context_sync(self.ctx))

    def last_groupby_path(self):
        """'dense' | 'hash' | 'sort' | None: the path that served the last GROUP BY entry (diagnostic, include/hark.h)."""
        return {1: "dense", 2: "hash", 3: "sort", 4: "small"}.get(self.lib.hark_context_last_groupby_path(self.ctx))

    def last_groupby_passes(self):
        """Passes over the table's rows of the last dense-path filter_groupby (diagnostic, include/hark.h)."""
        return int(self.lib.hark_context_last_groupby_passes(self.ctx))

    def last_groupby_window(self):
        """True when the last dense GROUP BY found its key column sorted / clustered by the key and took the window kernels."""
        return self.lib.hark_context_last_groupby_window(self.ctx) == 1

    def last_groupby_rotated(self):
        """True when the last dense GROUP BY found neighbouring rows in one bucket (sorted files one behind the other) and ran the
        partition with rotated loads."""
        return self.lib.hark_context_last_groupby_window(self.ctx) == 2

    def last_join_path(self):
        """'partitioned' | 'sort-merge' | None: the path that served the last join entry (diagnostic, include/hark.h)."""
        return {1: "partitioned", 2: "sort-merge", 3: "partitioned, buckets cut by weight", 4: "clustered probe column, searched in row order", 5: "partitioned, rotated loads"}.get(self.lib.hark_context_last_join_path(self.ctx))

    def set_stream(self, raw_stream):
        """Run later entries on this hipStream_t handle; 0 / None = the context's own stream
        (so torch's DEFAULT stream, handle 0, cannot be shared: use dist.share_stream)."""
        self._chk(self.lib.hark_context_set_stream(self.ctx, raw_stream))

    def close(self):
        if self.ctx is not None:
            self.lib.hark_context_free(self.ctx)
            self.ctx = None

    # -- raw device memory -------------------------------------------------------
    def alloc(self, nbytes):
        p = C.c_void_p()
        self._chk(self.lib.hark_dev_alloc(self.ctx, C.byref(p), int(nbytes)))
        return p.value

    def free(self, ptr):
        self._chk(self.lib.hark_dev_free(self.ctx, ptr))

    def upload(self, ptr, arr):
        arr = np.ascontiguousarray(arr)
        self._chk(self.lib.hark_dev_upload(self.ctx, ptr, arr.ctypes.data, arr.nbytes))

    def download(self, ptr, n, dtype):
        """n elements at device address ptr as a numpy array; from 64 KiB on the array lives in a pinned block the copy engine
        wrote directly (tools/ingest_bench.py: the bounce-buffer path of earlier rounds reached 10-24 GB/s)."""
        dtype = np.dtype(dtype)
        nbytes = int(n) * dtype.itemsize
        if PINNED_FROM <= nbytes <= PINNED_UPTO:
            blk = C.c_void_p()
            self._chk(self.lib.hark_dev_download_pinned(self.ctx, ptr, nbytes, C.byref(blk)))
            return PinnedBlock(self, blk.value, nbytes).array(0, int(n), dtype)
        out = np.empty(n, dtype=dtype)
        self._chk(self.lib.hark_dev_download(self.ctx, out.ctypes.data, ptr, out.nbytes))
        return out

    def zero(self, ptr, nbytes):
        self._chk(self.lib.hark_op_zero(self.ctx, ptr, int(nbytes)))

    def gen_columns(self, seed, first_row, n, G, exact, p=None, k=None, v=None):
        self._chk(self.lib.hark_op_gen_columns(self.ctx, int(seed), int(first_row), int(n), int(G), 1 if exact else 0, p, k, v))

    def stream_read(self, ptrs, nbytes_each, fold_ptr):
        """Read 1..3 device buffers of nbytes_each bytes once, concurrently (bandwidth probe); XORs a
        64-bit fold of everything read into the device u64 at fold_ptr."""
        ptrs = [ptrs] if isinstance(ptrs, int) else list(ptrs)
        arr = (C.c_void_p * len(ptrs))(*ptrs)
        self._chk(self.lib.hark_op_stream_read(self.ctx, arr, len(ptrs), int(nbytes_each), fold_ptr))

    def stream_mix(self, ptrs, nbytes_each, dst_ptr, sixteenths=12):
        """Read three device buffers once and write sixteenths/16 of one buffer's volume to dst_ptr (traffic-mix probe)."""
        arr = (C.c_void_p * 3)(*list(ptrs))
        self._chk(self.lib.hark_op_stream_mix(self.ctx, arr, int(nbytes_each), dst_ptr, int(sixteenths)))

    def partition_by_hash(self, key_ptr, dtype, n, nparts, perm_ptr):
        """Row ids grouped by hash(key) part into perm_ptr (device u32[n]); returns the part sizes."""
        counts = (C.c_int64 * int(nparts))()
        self._chk(self.lib.hark_op_partition_by_hash(self.ctx, key_ptr, _ffi.DT_OF[np.dtype(dtype)], int(n), int(nparts), perm_ptr, counts))
        return list(counts)

    def partition_by_range(self, key_ptr, dtype, n, splitters, descending, perm_ptr):
        """Row ids grouped by key range (part = number of splitters <= key, mirrored when
        descending) into perm_ptr (device u32[n]); returns the len(splitters)+1 part sizes."""
        sp = np.ascontiguousarray(splitters, dtype=np.dtype(dtype))
        counts = (C.c_int64 * (sp.size + 1))()
        self._chk(self.lib.hark_op_partition_by_range(self.ctx, key_ptr, _ffi.DT_OF[np.dtype(dtype)], int(n), sp.size + 1,
                                                      sp.ctypes.data_as(C.c_void_p), 1 if descending else 0, perm_ptr, counts))
        return list(counts)

    def gather(self, src_ptr, dtype, idx_ptr, dst_ptr, n):
        self._chk(self.lib.hark_op_gather(self.ctx, src_ptr, _ffi.DT_OF[np.dtype(dtype)], idx_ptr, dst_ptr, int(n)))

    # -- tables ------------------------------------------------------------------
    def table_from_matrix(self, mat, dtype):
        """Upload an [n][m] host matrix as m device columns of `dtype`
        (futhark_new_i32_2d / _u32_2d).  Values are converted with numpy
        wrap-around semantics first (the reference hands int64 to an i32/u32
        entry)."""
        dtype = np.dtype(dtype)
        a = np.asarray(mat)
        if a.ndim != 2:
            a = a.reshape(0, 0) if a.size == 0 else a.reshape(a.shape[0], -1)
        if a.dtype != dtype:
            a = a.astype(dtype)            # keeps F order for F-ordered input
        if not (a.flags.c_contiguous or a.flags.f_contiguous):
            a = np.ascontiguousarray(a)
        n, m = a.shape
        es = a.itemsize
        rs, cs = (a.strides[0] // es, a.strides[1] // es) if n and m else (m, 1)
        h = C.c_void_p()
        self._chk(self.lib.hark_table_new_2d(self.ctx, C.byref(h), a.ctypes.data, _ffi.DT_OF[dtype], n, m, rs, cs))
        return DeviceTable(self, h)

    def table_from_columns(self, cols):
        cols = [np.ascontiguousarray(c) for c in cols]
        n = len(cols[0]) if cols else 0
        dts = (C.c_int32 * len(cols))(*[_ffi.DT_OF[c.dtype] for c in cols])
        ptrs = (C.c_void_p * len(cols))(*[c.ctypes.data for c in cols])
        h = C.c_void_p()
        self._chk(self.lib.hark_table_new_columns(self.ctx, C.byref(h), n, len(cols), dts, ptrs))
        return DeviceTable(self, h)

    def table_from_device(self, n, ptrs, dtypes, keepalive=None):
        dts = (C.c_int32 * len(ptrs))(*[_ffi.DT_OF[np.dtype(d)] for d in dtypes])
        pp = (C.c_void_p * len(ptrs))(*ptrs)
        h = C.c_void_p()
        self._chk(self.lib.hark_table_from_device(self.ctx, C.byref(h), int(n), len(ptrs), dts, pp))
        return DeviceTable(self, h, keepalive)

    # -- entries -------------------------------------------------------------------
    def query_sel(self, table, cols):
        a, p = _ffi.i32_array(cols)
        h = C.c_void_p()
        self._chk(self.lib.hark_entry_query_sel(self.ctx, C.byref(h), table._h, p, a.size))
        return Result(self, h)

    def query_groupby(self, table, g_col, s_cols, t_cols):
        a, pa = _ffi.i32_array(s_cols)
        b, pb = _ffi.i32_array(t_cols)
        h = C.c_void_p()
        self._chk(self.lib.hark_entry_query_groupby(self.ctx, C.byref(h), table._h, int(g_col), pa, a.size, pb, b.size))
        return Result(self, h)

    def join(self, t1, t2, col1, col2, cols1, cols2):
        a, pa = _ffi.i32_array(cols1)
        b, pb = _ffi.i32_array(cols2)
        h = C.c_void_p()
        self._chk(self.lib.hark_entry_join(self.ctx, C.byref(h), t1._h, t2._h, int(col1), int(col2), pa, a.size, pb, b.size))
        return Result(self, h)

    def _const(self, table, col, value, cmp=">"):
        """(comparison, constant in the column's dtype) equivalent to `column <cmp> value` over the column's values."""
        return normalise_predicate(np.dtype(table.dtype(col)), cmp, value)

    def _predicates(self, table, where):
        """where = [(col, cmp, value), ...] -> ctypes arrays (columns, comparison opcodes, constant pointers) + keepalive.
        Every literal is normalised to its column's dtype first (normalise_predicate)."""
        consts, cols, cmps, ptrs = [], [], [], []
        for col, cmp, value in where:
            if cmp == "mask":                                      # a predicate tree evaluated already (predicate_tree_mask): value = the DeviceBuffer
                consts.append(value); cols.append(0); cmps.append(_ffi.CMP["mask"]); ptrs.append(value.ptr)
                continue
            cmp2, c = self._const(table, col, value, cmp)
            consts.append(c)
            cols.append(int(col))
            cmps.append(_ffi.CMP[cmp2])
            ptrs.append(c.ctypes.data)
        n = len(where)
        return ((C.c_int32 * n)(*cols), (C.c_int32 * n)(*cmps), (C.c_void_p * n)(*ptrs), consts)

    def column_expr(self, table, node):
        """An arithmetic expression over the table's columns -> (DeviceBuffer, numpy dtype): node = ("col", j) | ("num", v) |
        ("add" | "sub" | "mul" | "div", left, right).  One elementwise kernel per operator (hark_op_column_binary): integers stay
        integers (i32 when both operands are 32-bit: wrapping; i64 otherwise), a division or a float operand makes f32."""
        n = table.shape[0]
        ops = {"add": 0, "sub": 1, "mul": 2, "div": 3}

        def ev(nd):
            """-> (buffer | None, device pointer | None, dtype | None, constant)."""
            if nd[0] == "col":
                return None, table.device_ptr(nd[1]), np.dtype(table.dtype(nd[1])), 0.0
            if nd[0] == "num":
                return None, None, None, float(nd[1])
            (_, px, dx, cx), (_, py, dy, cy) = kx, ky = ev(nd[1]), ev(nd[2])
            if px is None and py is None:                          # two constants: folded here
                v = {"add": cx + cy, "sub": cx - cy, "mul": cx * cy, "div": cx / cy if cy else float("nan")}[nd[0]]
                return None, None, None, v
            integral = lambda d, c: (d is not None and d.kind in "iu") or (d is None and float(c).is_integer())
            if nd[0] != "div" and integral(dx, cx) and integral(dy, cy):
                wide = any(d is not None and (d.itemsize == 8 or d == np.dtype(np.uint32)) for d in (dx, dy)) or any(d is None and not -2**31 <= c < 2**31 for d, c in ((dx, cx), (dy, cy)))
                out = np.dtype(np.int64 if wide else np.int32)
            else:
                out = np.dtype(np.float32)
            buf = DeviceBuffer(self, self.alloc(max(n, 1) * out.itemsize))
            code = lambda d: 0 if d is None else _ffi.DT_OF[d]
            self._chk(self.lib.hark_op_column_binary(self.ctx, n, ops[nd[0]], px, code(dx), cx, py, code(dy), cy, _ffi.DT_OF[out], buf.ptr))
            buf._keep = (kx[0], ky[0])                             # operand buffers live until the kernel has run (stream order: freed blocks are reused in order)
            return buf, buf.ptr, out, 0.0

        buf, ptr, dt, c = ev(node)
        if buf is None:                                            # a bare column or a constant: materialise as column + 0 / 0 + c
            return self.column_expr(table, ("add", node, ("num", 0)))
        return buf, dt

    def predicate_tree_mask(self, table, node):
        """A WHERE tree -> survivor bitmask on the device (hark_op_predicate_tree), as a DeviceBuffer to pass on as the conjunct
        (None, "mask", buffer).  node = ("and" | "or", [nodes]) | ("not", node) | ("cmp", col, cmp, number) |
        ("cmpcol", col, cmp, col2) | ("in", col, [numbers]).  Literals are normalised to the column's dtype like every
        predicate's (normalise_predicate); IN is the OR of its equalities."""
        kinds, a, b, consts = [], [], [], []

        def leaf(col, cmp, value):
            cmp2, c = self._const(table, col, value, cmp)
            kinds.append(0); a.append(int(col)); b.append(_ffi.CMP[cmp2]); consts.append(c)

        def emit(nd):
            tag = nd[0]
            if tag in ("and", "or"):
                kids = list(nd[1])
                if not kids:
                    raise _ffi.HarkError(_ffi.EARG, "predicate tree: empty " + tag)
                emit(kids[0])
                for k in kids[1:]:
                    emit(k)
                    kinds.append(2 if tag == "and" else 3); a.append(0); b.append(0); consts.append(None)
            elif tag == "not":
                emit(nd[1])
                kinds.append(4); a.append(0); b.append(0); consts.append(None)
            elif tag == "cmp":
                leaf(nd[1], nd[2], nd[3])
            elif tag == "cmpcol":
                if np.dtype(table.dtype(nd[1])) != np.dtype(table.dtype(nd[3])):
                    raise _ffi.HarkError(_ffi.EUNSUPPORTED, "a comparison of two columns needs columns of one dtype")
                kinds.append(1); a.append(int(nd[1])); b.append(_ffi.CMP[_CANON.get(nd[2], nd[2])] | (int(nd[3]) << 4)); consts.append(None)
            elif tag == "in":
                vals = list(nd[2])
                if not vals:
                    raise _ffi.HarkError(_ffi.EARG, "IN with an empty list")
                leaf(nd[1], "=", vals[0])
                for v in vals[1:]:
                    leaf(nd[1], "=", v)
                    kinds.append(3); a.append(0); b.append(0); consts.append(None)
            else:
                raise _ffi.HarkError(_ffi.EARG, f"predicate tree: unknown node {tag!r}")

        emit(node)
        n = len(kinds)
        rows = table.shape[0]
        buf = DeviceBuffer(self, self.alloc((rows + 7) // 8 + 16))
        if rows:
            self._chk(self.lib.hark_op_predicate_tree(self.ctx, table._h, n, (C.c_int32 * n)(*kinds), (C.c_int32 * n)(*a), (C.c_int32 * n)(*b),
                                                      (C.c_void_p * n)(*[None if c is None else c.ctypes.data for c in consts]), buf.ptr))
        return buf

    def filter_sel(self, table, where_col, cmp=None, value=None, cols=(), want_row_index=True):
        """WHERE + projection.  `where_col` is a column index (with cmp, value) or an AND-list [(col, cmp, value), ...]:
        all conjuncts are evaluated into one survivor mask, no intermediate table is materialised."""
        where = list(where_col) if isinstance(where_col, (list, tuple)) else [(where_col, cmp, value)]
        a, pa = _ffi.i32_array(cols)
        wc, wo, wp, keep = self._predicates(table, where)
        h = C.c_void_p()
        self._chk(self.lib.hark_entry_filter_sel_and(self.ctx, C.byref(h), table._h, len(where), wc, wo, wp, pa, a.size, 1 if want_row_index else 0))
        return Result(self, h)

    def filter_groupby(self, table, where, g_col, aggs):
        """where = None | (col, cmp, value) | [(col, cmp, value), ...] (AND); aggs = [(op_name, col), ...]."""
        cols, pc = _ffi.i32_array([c for _, c in aggs])
        ops, po = _ffi.i32_array([_ffi.AGG[o] for o, _ in aggs])
        if where is None:
            where = []
        elif where and not isinstance(where[0], (list, tuple)):
            where = [tuple(where)]
        wc, wo, wp, keep = self._predicates(table, list(where))
        h = C.c_void_p()
        self._chk(self.lib.hark_entry_filter_groupby_and(self.ctx, C.byref(h), table._h, len(where), wc, wo, wp, int(g_col), pc, po, cols.size))
        return Result(self, h)

    def topk(self, table, where, key_col, descending, k, cols):
        """The first k (<= 64) rows of filter_sel(where) + sort(key_col) without materialising either."""
        a, pa = _ffi.i32_array(cols)
        where = list(where or [])
        wc, wo, wp, keep = self._predicates(table, where)
        h = C.c_void_p()
        self._chk(self.lib.hark_entry_topk(self.ctx, C.byref(h), table._h, len(where), wc, wo, wp, int(key_col), 1 if descending else 0, int(k), pa, a.size))
        return Result(self, h)

    @staticmethod
    def agg_result_dtype(op, col_dtype):
        """dtype of an aggregate's result column (hark_entry_filter_groupby*, include/hark.h)."""
        if op == "count":
            return np.dtype(np.int64)
        if op == "avg":
            return np.dtype(np.float32)
        if op == "sum":
            return np.dtype(np.float32) if np.dtype(col_dtype) == np.float32 else np.dtype(np.int64)
        return np.dtype(col_dtype)

    def filter_groupby_topk(self, table, where, g_col, aggs, having, order_item, descending, k):
        """WHERE + GROUP BY + HAVING + ORDER BY + LIMIT k in one call (dense keys): Result [key, aggregates...] with at most
        k rows, or None when the entry declines (keys not dense, k too large) -- compose filter_groupby + topk then.
        having = [(item, cmp, value), ...]; items: 0 = the key, j + 1 = aggregate j."""
        cols, pc = _ffi.i32_array([c for _, c in aggs])
        ops, po = _ffi.i32_array([_ffi.AGG[o] for o, _ in aggs])
        where = list(where or [])
        wc, wo, wp, keep = self._predicates(table, where)
        item_dtype = lambda it: np.dtype(table.dtype(g_col)) if it == 0 else self.agg_result_dtype(aggs[it - 1][0], table.dtype(aggs[it - 1][1]))
        hconsts, hitems, hcmps = [], [], []
        for it, cmp, value in having:
            cmp2, c = normalise_predicate(item_dtype(it), cmp, value)
            hconsts.append(c); hitems.append(int(it)); hcmps.append(_ffi.CMP[cmp2])
        nh = len(hitems)
        h = C.c_void_p()
        rc = self.lib.hark_entry_filter_groupby_topk(self.ctx, C.byref(h), table._h, len(where), wc, wo, wp, int(g_col), pc, po, cols.size,
                                                     nh, (C.c_int32 * nh)(*hitems), (C.c_int32 * nh)(*hcmps), (C.c_void_p * nh)(*[c.ctypes.data for c in hconsts]),
                                                     int(order_item), 1 if descending else 0, int(k))
        if rc == _ffi.EUNSUPPORTED:
            return None
        self._chk(rc)
        return Result(self, h)

    def filter_groupby_slots(self, table, where, g_col, G, aggs):
        """Partial aggregates of this table over the dense key domain [0, G), one row per key slot: Result
        [aggregate..., COUNT(*)] with G rows (sum / min / max / count only), or None when the shape is not the fused dense
        one.  What a shard contributes to an all-reduce merge (dist.ShardedFutharkContext)."""
        cols, pc = _ffi.i32_array([c for _, c in aggs])
        ops, po = _ffi.i32_array([_ffi.AGG[o] for o, _ in aggs])
        where = list(where or [])
        wc, wo, wp, keep = self._predicates(table, where)
        h = C.c_void_p()
        rc = self.lib.hark_entry_filter_groupby_slots(self.ctx, C.byref(h), table._h, len(where), wc, wo, wp, int(g_col), int(G), pc, po, cols.size)
        if rc == _ffi.EUNSUPPORTED:
            return None
        self._chk(rc)
        return Result(self, h)

    def filter_groupby_subset(self, table, where, g_col, keys, aggs):
        """The aggregates `aggs` for the groups `keys` only (a numpy array of <= 1024 distinct 32-bit key values): a Result
        with len(keys) rows in that order, one column per aggregate.  Raises on unsupported shapes (see include/hark.h)."""
        cols, pc = _ffi.i32_array([c for _, c in aggs])
        ops, po = _ffi.i32_array([_ffi.AGG[o] for o, _ in aggs])
        where = list(where or [])
        wc, wo, wp, keep = self._predicates(table, where)
        k = np.ascontiguousarray(keys).view(np.uint32) if np.asarray(keys).dtype.itemsize == 4 else np.ascontiguousarray(keys, dtype=np.uint32)
        h = C.c_void_p()
        self._chk(self.lib.hark_entry_filter_groupby_subset(self.ctx, C.byref(h), table._h, len(where), wc, wo, wp, int(g_col), k.ctypes.data, k.size, pc, po, cols.size))
        return Result(self, h)

    def predicate_bitmask(self, table, where, mask_ptr):
        """AND of the predicates as a survivor bitmask (bit r & 7 of byte r >> 3) at device address mask_ptr; feed it to
        FgbPlan.run(p=mask_ptr, cmp="mask")."""
        wc, wo, wp, keep = self._predicates(table, list(where))
        self._chk(self.lib.hark_op_predicate_bitmask(self.ctx, table._h, len(where), wc, wo, wp, mask_ptr))

    def composite_key(self, table, key_cols, ranges=None):
        """Fold several 32-bit integer key columns into one composite key column on the device.
        ranges = None (use the table's own column statistics) or (mins, spans) to encode with (sharded tables:
        the ranges over all shards).  Returns (DeviceBuffer owning the column, numpy dtype, mins, spans)."""
        a, pa = _ffi.i32_array(key_cols)
        out, dt = C.c_void_p(), C.c_int32()
        mins, spans = (C.c_int64 * a.size)(), (C.c_int64 * a.size)()
        if ranges is not None:
            for j in range(a.size):
                mins[j], spans[j] = int(ranges[0][j]), int(ranges[1][j])
        self._chk(self.lib.hark_table_composite_key(self.ctx, table._h, pa, a.size, C.byref(out), C.byref(dt), mins, spans,
                                                    0 if ranges is None else 1))
        return DeviceBuffer(self, out.value), _ffi.NP_OF[dt.value], list(mins), list(spans)

    def column_range(self, table, col):
        """(min, max) of a 32-bit integer column of a non-empty device table."""
        lo, hi = C.c_int64(), C.c_int64()
        self._chk(self.lib.hark_table_column_range(self.ctx, table._h, int(col), C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def sort(self, table, key_col, cols, descending=False):
        a, pa = _ffi.i32_array(cols)
        h = C.c_void_p()
        self._chk(self.lib.hark_entry_sort(self.ctx, C.byref(h), table._h, int(key_col), 1 if descending else 0, pa, a.size))
        return Result(self, h)
