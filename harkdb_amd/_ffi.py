"""ctypes binding of libhark.so (include/hark.h).

This is the seam the reference fills with `futhark_ffi.Futhark(_main)`
(FutharkContext.py:31-41).  There is NO CPU fallback: if the HIP library is
missing or no GPU is visible the import / context creation raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HARK_LIB") or os.path.join(_HERE, "libhark.so")    # HARK_LIB: A/B builds on one box

OK, EBOUNDS, ENOMEM, EARG, EHIP, EUNSUPPORTED = range(6)
I32, U32, F32, I64 = range(4)
F64 = 4                                   # element type of a result MATRIX only (hark_result_matrix_pinned)
NP_OF = {I32: np.int32, U32: np.uint32, F32: np.float32, I64: np.int64}
DT_OF = {np.dtype(np.int32): I32, np.dtype(np.uint32): U32, np.dtype(np.float32): F32, np.dtype(np.int64): I64}
CMP = {">": 0, ">=": 1, "<": 2, "<=": 3, "=": 4, "==": 4, "!=": 5, "<>": 5, "mask": 6}
AGG = {"key": 0, "prod": 1, "sum": 2, "max": 3, "min": 4, "count": 5, "avg": 6}


class HarkError(Exception):
    """Raised for every non-zero status from libhark.so (the reference raises
    plain Exception for its own failures, parse.py:33 etc.)."""

    def __init__(self, code, msg):
        super().__init__(f"[hark {code}] {msg}")
        self.code = code


_vp, _i32, _i64, _u64, _u32, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_uint32, C.c_float
_pp = C.POINTER(C.c_void_p)

# name -> (restype, argtypes); every symbol include/hark.h declares.
SIGNATURES = {
    "hark_version": (C.c_int, []),
    "hark_context_new": (C.c_int, [_pp, C.c_int]),
    "hark_context_free": (None, [_vp]),
    "hark_context_sync": (C.c_int, [_vp]),
    "hark_context_get_error": (C.c_char_p, [_vp]),
    "hark_context_set_stream": (C.c_int, [_vp, _vp]),
    "hark_context_trim": (C.c_int, [_vp]),
    "hark_table_new_2d": (C.c_int, [_vp, _pp, _vp, C.c_int, _i64, _i64, _i64, _i64]),
    "hark_table_new_columns": (C.c_int, [_vp, _pp, _i64, _i64, C.POINTER(_i32), _pp]),
    "hark_table_from_device": (C.c_int, [_vp, _pp, _i64, _i64, C.POINTER(_i32), _pp]),
    "hark_table_shape": (C.c_int, [_vp, C.POINTER(_i64), C.POINTER(_i64)]),
    "hark_table_dtype": (C.c_int, [_vp, _i64]),
    "hark_table_column_device": (_vp, [_vp, _i64]),
    "hark_table_free": (C.c_int, [_vp, _vp]),
    "hark_result_shape": (C.c_int, [_vp, C.POINTER(_i64), C.POINTER(_i64)]),
    "hark_result_dtype": (C.c_int, [_vp, _i64]),
    "hark_result_values_2d": (C.c_int, [_vp, _vp, _vp, C.c_int]),
    "hark_result_column": (C.c_int, [_vp, _vp, _i64, _vp]),
    "hark_result_column_device": (_vp, [_vp, _i64]),
    "hark_result_columns_prefix": (C.c_int, [_vp, _vp, _i64, _vp]),
    "hark_host_alloc": (C.c_int, [_vp, _pp, C.c_size_t]),
    "hark_host_free": (C.c_int, [_vp, _vp]),
    "hark_result_columns_pinned": (C.c_int, [_vp, _vp, _i64, _pp, C.POINTER(_i64)]),
    "hark_result_matrix_pinned": (C.c_int, [_vp, _vp, C.POINTER(_i32), _i64, _i64, C.c_int, _pp]),
    "hark_dev_download_pinned": (C.c_int, [_vp, _vp, C.c_size_t, _pp]),
    "hark_result_free": (C.c_int, [_vp, _vp]),
    "hark_entry_query_sel": (C.c_int, [_vp, _pp, _vp, C.POINTER(_i32), _i64]),
    "hark_entry_query_groupby": (C.c_int, [_vp, _pp, _vp, _i32, C.POINTER(_i32), _i64, C.POINTER(_i32), _i64]),
    "hark_entry_join": (C.c_int, [_vp, _pp, _vp, _vp, _i32, _i32, C.POINTER(_i32), _i64, C.POINTER(_i32), _i64]),
    "hark_entry_filter_sel": (C.c_int, [_vp, _pp, _vp, _i32, _i32, _vp, C.POINTER(_i32), _i64, _i32]),
    "hark_entry_filter_sel_and": (C.c_int, [_vp, _pp, _vp, _i64, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(C.c_void_p), C.POINTER(_i32), _i64, _i32]),
    "hark_entry_filter_groupby_and": (C.c_int, [_vp, _pp, _vp, _i64, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(C.c_void_p), _i32, C.POINTER(_i32), C.POINTER(_i32), _i64]),
    "hark_op_predicate_bitmask": (C.c_int, [_vp, _vp, _i64, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(C.c_void_p), _vp]),
    "hark_op_column_binary": (C.c_int, [_vp, _i64, _i32, _vp, _i32, C.c_double, _vp, _i32, C.c_double, _i32, _vp]),
    "hark_op_predicate_tree": (C.c_int, [_vp, _vp, _i64, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32), C.POINTER(C.c_void_p), _vp]),
    "hark_entry_filter_groupby": (C.c_int, [_vp, _pp, _vp, _i32, _i32, _vp, _i32, C.POINTER(_i32), C.POINTER(_i32), _i64]),
    "hark_entry_topk": (C.c_int, [_vp, _pp, _vp, _i64, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(C.c_void_p), _i32, _i32, _i64, C.POINTER(_i32), _i64]),
    "hark_entry_filter_groupby_topk": (C.c_int, [_vp, _pp, _vp, _i64, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(C.c_void_p), _i32, C.POINTER(_i32), C.POINTER(_i32), _i64,
                                                 _i64, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(C.c_void_p), _i32, _i32, _i64]),
    "hark_entry_filter_groupby_slots": (C.c_int, [_vp, _pp, _vp, _i64, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(C.c_void_p), _i32, _i64, C.POINTER(_i32), C.POINTER(_i32), _i64]),
    "hark_entry_filter_groupby_subset": (C.c_int, [_vp, _pp, _vp, _i64, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(C.c_void_p), _i32, _vp, _i64, C.POINTER(_i32), C.POINTER(_i32), _i64]),
    "hark_entry_sort": (C.c_int, [_vp, _pp, _vp, _i32, _i32, C.POINTER(_i32), _i64]),
    "hark_op_segmented_scan_add_i32": (C.c_int, [_vp, _vp, _vp, _i64, _vp]),
    "hark_op_segmented_reduce_add_i32": (C.c_int, [_vp, _vp, _vp, _i64, _vp, C.POINTER(_i64)]),
    "hark_op_replicated_iota": (C.c_int, [_vp, _vp, _i64, _vp, C.POINTER(_i64)]),
    "hark_op_segmented_iota": (C.c_int, [_vp, _vp, _i64, _vp]),
    "hark_op_expand_indices": (C.c_int, [_vp, _vp, _i64, _vp, _vp, C.POINTER(_i64)]),
    "hark_op_partition_by_hash": (C.c_int, [_vp, _vp, _i32, _i64, _i32, _vp, C.POINTER(_i64)]),
    "hark_table_composite_key": (C.c_int, [_vp, _vp, C.POINTER(_i32), _i64, C.POINTER(C.c_void_p), C.POINTER(_i32), C.POINTER(_i64), C.POINTER(_i64), _i32]),
    "hark_table_column_range": (C.c_int, [_vp, _vp, _i32, C.POINTER(_i64), C.POINTER(_i64)]),
    "hark_table_invalidate_stats": (C.c_int, [_vp, _vp, _i32]),
    "hark_context_last_groupby_path": (C.c_int, [_vp]),
    "hark_context_last_groupby_passes": (C.c_int, [_vp]),
    "hark_context_last_groupby_window": (C.c_int, [_vp]),
    "hark_context_last_join_path": (C.c_int, [_vp]),
    "hark_op_stream_read": (C.c_int, [_vp, C.POINTER(C.c_void_p), _i32, _i64, _vp]),
    "hark_op_stream_mix": (C.c_int, [_vp, C.POINTER(C.c_void_p), _i64, _vp, _i32]),
    "hark_op_partition_by_range": (C.c_int, [_vp, _vp, _i32, _i64, _i32, _vp, _i32, _vp, C.POINTER(_i64)]),
    "hark_op_gather": (C.c_int, [_vp, _vp, _i32, _vp, _vp, _i64]),
    "hark_op_gen_columns": (C.c_int, [_vp, _u64, _i64, _i64, _u32, _i32, _vp, _vp, _vp]),
    "hark_fgb_plan_new": (C.c_int, [_vp, _pp, _i64, _i64]),
    "hark_fgb_plan_free": (C.c_int, [_vp, _vp]),
    "hark_fgb_plan_set": (C.c_int, [_vp, C.c_char_p, _i64]),
    "hark_fgb_reset": (C.c_int, [_vp, _vp]),
    "hark_op_filter_groupby_dense_f32": (C.c_int, [_vp, _vp, _vp, _i32, _f32, _vp, _vp, _i64]),
    "hark_fgb_acc_device": (C.c_int, [_vp, _pp, _pp]),
    "hark_fgb_finish": (C.c_int, [_vp, _vp, _vp, _vp]),
    "hark_fgb_finish_async": (C.c_int, [_vp, _vp, _vp, _vp]),
    "hark_fgb_check": (C.c_int, [_vp, _vp]),
    "hark_op_groupby_dense_u32": (C.c_int, [_vp, _vp, _vp, _vp, _i64]),
    "hark_fgb_finish_u32": (C.c_int, [_vp, _vp, _vp, _vp]),
    "hark_fgb_timing": (C.c_int, [_vp, _vp, C.POINTER(C.c_double), C.POINTER(_i64)]),
    "hark_op_zero": (C.c_int, [_vp, _vp, _i64]),
    "hark_dev_alloc": (C.c_int, [_vp, _pp, _i64]),
    "hark_dev_free": (C.c_int, [_vp, _vp]),
    "hark_dev_upload": (C.c_int, [_vp, _vp, _vp, _i64]),
    "hark_dev_download": (C.c_int, [_vp, _vp, _vp, _i64]),
}

# The generated-Futhark-C-API names (include/futhark_compat.h): what `futhark_ffi.Futhark(_main)` binds in the
# reference (FutharkContext.py:31-41).  The package itself goes through the hark_* names above; these are bound for
# callers (and tests) that speak the reference's FFI.
FUTHARK_SIGNATURES = {
    "futhark_context_config_new": (_vp, []),
    "futhark_context_config_free": (None, [_vp]),
    "futhark_context_config_set_debugging": (None, [_vp, C.c_int]),
    "futhark_context_config_set_profiling": (None, [_vp, C.c_int]),
    "futhark_context_config_set_logging": (None, [_vp, C.c_int]),
    "futhark_context_config_set_device": (None, [_vp, C.c_char_p]),
    "futhark_context_new": (_vp, [_vp]),
    "futhark_context_free": (None, [_vp]),
    "futhark_context_sync": (C.c_int, [_vp]),
    "futhark_context_get_error": (_vp, [_vp]),               # malloc'd char*: the caller frees it
    "futhark_context_report": (_vp, [_vp]),
    "futhark_context_clear_caches": (C.c_int, [_vp]),
    "futhark_context_pause_profiling": (None, [_vp]),
    "futhark_context_unpause_profiling": (None, [_vp]),
    "futhark_new_i32_1d": (_vp, [_vp, _vp, _i64]),
    "futhark_free_i32_1d": (C.c_int, [_vp, _vp]),
    "futhark_values_i32_1d": (C.c_int, [_vp, _vp, _vp]),
    "futhark_shape_i32_1d": (C.POINTER(_i64), [_vp, _vp]),
    "futhark_new_i32_2d": (_vp, [_vp, _vp, _i64, _i64]),
    "futhark_free_i32_2d": (C.c_int, [_vp, _vp]),
    "futhark_values_i32_2d": (C.c_int, [_vp, _vp, _vp]),
    "futhark_shape_i32_2d": (C.POINTER(_i64), [_vp, _vp]),
    "futhark_new_u32_2d": (_vp, [_vp, _vp, _i64, _i64]),
    "futhark_free_u32_2d": (C.c_int, [_vp, _vp]),
    "futhark_values_u32_2d": (C.c_int, [_vp, _vp, _vp]),
    "futhark_shape_u32_2d": (C.POINTER(_i64), [_vp, _vp]),
    "futhark_entry_query_sel": (C.c_int, [_vp, _pp, _vp, _vp]),
    "futhark_entry_query_groupby": (C.c_int, [_vp, _pp, _vp, _i32, _vp, _vp]),
    "futhark_entry_join": (C.c_int, [_vp, _pp, _vp, _vp, _i32, _i32, _vp, _vp]),
}


def bind_futhark_names(lib):
    for name, (res, args) in FUTHARK_SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


_lib = None


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so.  If libhark.so pulls in the
    system copy first, a later `import torch` in the same process fails to find a GPU
    ("No HIP GPUs are available", reproduced with libhark loaded before torch).  Loading
    torch's copy first (without importing torch) makes both sides use one runtime."""
    if os.environ.get("HARK_SYSTEM_HIP"):
        return
    import importlib.util
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """dlopen libhark.so and bind every declared symbol.  Raises ImportError
    with build instructions when the library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C harkdb_amd/csrc` "
            "(or python -c 'import __graft_entry__ as g; g.build()').  There is no CPU fallback.")
    _share_hip_runtime_with_torch()
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def i32_array(seq):
    a = np.ascontiguousarray(np.asarray(seq, dtype=np.int64).astype(np.int32))
    return a, a.ctypes.data_as(C.POINTER(_i32))
