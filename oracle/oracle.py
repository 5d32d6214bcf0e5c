"""ctypes wrapper around oracle/liboracle.so (the CPU restatement of HarkDB's
Futhark operators).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Nothing under harkdb_amd/ may import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

CMP = {">": 0, ">=": 1, "<": 2, "<=": 3, "=": 4, "==": 4, "!=": 5, "<>": 5}


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "hark_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.ora_free.argtypes = [C.c_void_p]
        _LIB.ora_free.restype = None
        for name in ("ora_filter_f32", "ora_filter_i32", "ora_filter_u32", "ora_filter_i64"):
            getattr(_LIB, name).restype = C.c_int64
    return _LIB


class OracleError(Exception):
    pass


def _p(a, ty):
    return a.ctypes.data_as(C.POINTER(ty))


def _check(rc):
    if rc != 0:
        raise OracleError({1: "bounds error", 2: "out of memory", 3: "bad argument"}.get(rc, f"rc={rc}"))


# ---- segmented.fut primitives (KAT surface) --------------------------------

def segmented_scan_add(flags, vals):
    f = np.ascontiguousarray(flags, dtype=np.uint8)
    a = np.ascontiguousarray(vals, dtype=np.int32)
    out = np.empty_like(a)
    _check(lib().ora_segmented_scan_add_i32(_p(f, C.c_uint8), _p(a, C.c_int32), C.c_int64(a.size), _p(out, C.c_int32)))
    return out


def segmented_reduce_add(flags, vals):
    f = np.ascontiguousarray(flags, dtype=np.uint8)
    a = np.ascontiguousarray(vals, dtype=np.int32)
    out = np.empty(max(a.size, 1), dtype=np.int32)
    n = C.c_int64(0)
    _check(lib().ora_segmented_reduce_add_i32(_p(f, C.c_uint8), _p(a, C.c_int32), C.c_int64(a.size),
                                              _p(out, C.c_int32), C.byref(n)))
    return out[: n.value].copy()


def _var_out(fn, arr, cap):
    a = np.ascontiguousarray(arr, dtype=np.int32)
    out = np.empty(max(cap, 1), dtype=np.int32)
    n = C.c_int64(0)
    _check(fn(_p(a, C.c_int32), C.c_int64(a.size), _p(out, C.c_int32), C.byref(n)))
    return out[: n.value].copy()


def replicated_iota(reps):
    reps = np.asarray(reps, dtype=np.int32)
    return _var_out(lib().ora_replicated_iota, reps, int(reps.sum()) if reps.size else 0)


def segmented_iota(flags):
    f = np.ascontiguousarray(flags, dtype=np.uint8)
    out = np.empty(f.size, dtype=np.int32)
    _check(lib().ora_segmented_iota(_p(f, C.c_uint8), C.c_int64(f.size), _p(out, C.c_int32)))
    return out


def test_expand(arr):
    arr = np.asarray(arr, dtype=np.int32)
    return _var_out(lib().ora_test_expand, arr, int(arr.sum()) if arr.size else 0)


def test_expand_reduce(arr):
    arr = np.asarray(arr, dtype=np.int32)
    return _var_out(lib().ora_test_expand_reduce, arr, arr.size)


def test_expand_outer_reduce(arr):
    arr = np.asarray(arr, dtype=np.int32)
    return _var_out(lib().ora_test_expand_outer_reduce, arr, arr.size)


# ---- operators --------------------------------------------------------------

def _mat(a, dtype):
    a = np.ascontiguousarray(np.asarray(a).astype(dtype, copy=False))
    if a.ndim != 2:
        a = a.reshape(0, 0) if a.size == 0 else a.reshape(a.shape[0], -1)
    return a


def query_sel(db, cols):
    """futhark/main.fut:7 -> select.fut:23."""
    db = _mat(np.asarray(db, dtype=np.int64).astype(np.int32), np.int32)
    cols = np.ascontiguousarray(cols, dtype=np.int32)
    n, m = db.shape
    out = np.empty((n, cols.size), dtype=np.int32)
    _check(lib().ora_query_sel(_p(db, C.c_int32), C.c_int64(n), C.c_int64(m), _p(cols, C.c_int32),
                               C.c_int64(cols.size), _p(out, C.c_int32)))
    return out


def query_groupby(db, g_col, s_cols, t_cols):
    """futhark/main.fut:9 -> groupby.fut:60-62."""
    db = _mat(np.asarray(db, dtype=np.int64).astype(np.uint32), np.uint32)
    s_cols = np.ascontiguousarray(s_cols, dtype=np.int32)
    t_cols = np.ascontiguousarray(t_cols, dtype=np.int32)
    n, m = db.shape
    res = C.POINTER(C.c_uint32)()
    g = C.c_int64(0)
    _check(lib().ora_query_groupby(_p(db, C.c_uint32), C.c_int64(n), C.c_int64(m), C.c_int32(int(g_col)),
                                   _p(s_cols, C.c_int32), C.c_int64(s_cols.size),
                                   _p(t_cols, C.c_int32), C.c_int64(t_cols.size), C.byref(res), C.byref(g)))
    w = s_cols.size + 1
    if g.value == 0:
        return np.empty((0, w), dtype=np.uint32)
    out = np.ctypeslib.as_array(res, shape=(g.value, w)).copy()
    lib().ora_free(res)
    return out


def join(db1, db2, col1, col2, cols1, cols2):
    """futhark/join.fut:52-75."""
    db1 = _mat(np.asarray(db1, dtype=np.int64).astype(np.uint32), np.uint32)
    db2 = _mat(np.asarray(db2, dtype=np.int64).astype(np.uint32), np.uint32)
    cols1 = np.ascontiguousarray(cols1, dtype=np.int32)
    cols2 = np.ascontiguousarray(cols2, dtype=np.int32)
    res = C.POINTER(C.c_uint32)()
    p = C.c_int64(0)
    _check(lib().ora_join(_p(db1, C.c_uint32), C.c_int64(db1.shape[0]), C.c_int64(db1.shape[1]),
                          _p(db2, C.c_uint32), C.c_int64(db2.shape[0]), C.c_int64(db2.shape[1]),
                          C.c_int32(int(col1)), C.c_int32(int(col2)),
                          _p(cols1, C.c_int32), C.c_int64(cols1.size), _p(cols2, C.c_int32), C.c_int64(cols2.size),
                          C.byref(res), C.byref(p)))
    w = cols1.size + cols2.size
    if p.value == 0 or w == 0:
        return np.empty((p.value if w == 0 else 0, w), dtype=np.uint32)
    out = np.ctypeslib.as_array(res, shape=(p.value, w)).copy()
    lib().ora_free(res)
    return out


# ---- extensions (parity unpinned by the reference) ---------------------------

def gen_columns(seed, first_row, n, G, exact=True, want=("p", "k", "v")):
    p = np.empty(n, dtype=np.float32) if "p" in want else None
    k = np.empty(n, dtype=np.int32) if "k" in want else None
    v = np.empty(n, dtype=np.float32) if "v" in want else None
    lib().ora_gen_columns(C.c_uint64(seed), C.c_int64(first_row), C.c_int64(n), C.c_uint32(G), C.c_int(1 if exact else 0),
                          _p(p, C.c_float) if p is not None else None,
                          _p(k, C.c_int32) if k is not None else None,
                          _p(v, C.c_float) if v is not None else None)
    return p, k, v


def filter_indices(col, op, c):
    col = np.ascontiguousarray(col)
    out = np.empty(col.size, dtype=np.int64)
    code = CMP[op]
    if col.dtype == np.float32:
        n = lib().ora_filter_f32(_p(col, C.c_float), C.c_int64(col.size), code, C.c_float(c), _p(out, C.c_int64))
    elif col.dtype == np.int32:
        n = lib().ora_filter_i32(_p(col, C.c_int32), C.c_int64(col.size), code, C.c_int32(int(c)), _p(out, C.c_int64))
    elif col.dtype == np.uint32:
        n = lib().ora_filter_u32(_p(col, C.c_uint32), C.c_int64(col.size), code, C.c_uint32(int(c)), _p(out, C.c_int64))
    elif col.dtype == np.int64:
        n = lib().ora_filter_i64(_p(col, C.c_int64), C.c_int64(col.size), code, C.c_int64(int(c)), _p(out, C.c_int64))
    else:
        raise TypeError(col.dtype)
    return out[:n].copy()


def filter_groupby_dense_f32(p, k, v, op, thr, G):
    """SELECT k, SUM(v), COUNT(*) WHERE p <op> thr GROUP BY k over keys in [0,G).
    Returns (sum32 sequential f32 fold, sum64, count int64)."""
    k = np.ascontiguousarray(k, dtype=np.int32)
    v = np.ascontiguousarray(v, dtype=np.float32)
    pp = None if p is None else np.ascontiguousarray(p, dtype=np.float32)
    s32 = np.empty(G, dtype=np.float32)
    s64 = np.empty(G, dtype=np.float64)
    cnt = np.empty(G, dtype=np.int64)
    _check(lib().ora_filter_groupby_dense_f32(_p(pp, C.c_float) if pp is not None else None, _p(k, C.c_int32),
                                              _p(v, C.c_float), C.c_int64(k.size), CMP[op], C.c_float(thr),
                                              C.c_int64(G), _p(s32, C.c_float), _p(s64, C.c_double), _p(cnt, C.c_int64)))
    return s32, s64, cnt


def filter_groupby_dense_f32_mt(p, k, v, op, thr, G, threads):
    """Multi-threaded single-pass direct-index aggregate (the stronger CPU baseline).  Returns (sum64, count)."""
    k = np.ascontiguousarray(k, dtype=np.int32)
    v = np.ascontiguousarray(v, dtype=np.float32)
    pp = None if p is None else np.ascontiguousarray(p, dtype=np.float32)
    s64 = np.empty(G, dtype=np.float64)
    cnt = np.empty(G, dtype=np.int64)
    _check(lib().ora_filter_groupby_dense_f32_mt(_p(pp, C.c_float) if pp is not None else None, _p(k, C.c_int32), _p(v, C.c_float),
                                                 C.c_int64(k.size), CMP[op], C.c_float(thr), C.c_int64(G), C.c_int(int(threads)),
                                                 _p(s64, C.c_double), _p(cnt, C.c_int64)))
    return s64, cnt


def filter_groupby_refalgo_f32(p, k, v, op, thr):
    """Same query through the reference's 32-pass sort + segmented fold."""
    k = np.ascontiguousarray(k, dtype=np.int32)
    v = np.ascontiguousarray(v, dtype=np.float32)
    pp = None if p is None else np.ascontiguousarray(p, dtype=np.float32)
    keys = C.POINTER(C.c_uint32)()
    sums = C.POINTER(C.c_float)()
    cnts = C.POINTER(C.c_uint32)()
    g = C.c_int64(0)
    _check(lib().ora_filter_groupby_refalgo_f32(_p(pp, C.c_float) if pp is not None else None, _p(k, C.c_int32),
                                                _p(v, C.c_float), C.c_int64(k.size), CMP[op], C.c_float(thr),
                                                C.byref(keys), C.byref(sums), C.byref(cnts), C.byref(g)))
    if g.value == 0:
        return (np.empty(0, np.uint32), np.empty(0, np.float32), np.empty(0, np.uint32))
    out = (np.ctypeslib.as_array(keys, shape=(g.value,)).copy(),
           np.ctypeslib.as_array(sums, shape=(g.value,)).copy(),
           np.ctypeslib.as_array(cnts, shape=(g.value,)).copy())
    for ptr in (keys, sums, cnts):
        lib().ora_free(ptr)
    return out


def argsort_u32(keys):
    keys = np.ascontiguousarray(keys, dtype=np.uint32)
    perm = np.empty(keys.size, dtype=np.int64)
    _check(lib().ora_argsort_u32(_p(keys, C.c_uint32), C.c_int64(keys.size), _p(perm, C.c_int64)))
    return perm
