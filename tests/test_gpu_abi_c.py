"""The drop-in boundary from C and under the generated Futhark API's own names.

* tests/abi_smoke.c (the program INTEGRATION.md shows) is compiled with gcc against include/ and libhark.so and run;
* G1/G2 (the reference's own two statements, README.md:42, test.py:7) are driven through ctypes using ONLY the
  futhark_* symbols `futhark c --library futhark/main.fut` would export (FutharkContext.py:41,65-66,70-71)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_golden, resolve_table

pytestmark = pytest.mark.gpu
OPS = load_golden("operators.json")


def test_compiled_c_caller(tmp_path):
    exe = str(tmp_path / "abi_smoke")
    subprocess.check_call(["gcc", "-O1", "-Wall", "-Werror", os.path.join(ROOT, "tests", "abi_smoke.c"), "-I", os.path.join(ROOT, "include"),
                           "-L", os.path.join(ROOT, "harkdb_amd"), "-lhark", "-Wl,-rpath," + os.path.join(ROOT, "harkdb_amd"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "abi_smoke ok" in out.stdout


@pytest.fixture(scope="module")
def fut():
    from harkdb_amd import _ffi
    lib = _ffi.load()
    _ffi.bind_futhark_names(lib)
    cfg = lib.futhark_context_config_new()
    ctx = lib.futhark_context_new(cfg)
    assert not lib.futhark_context_get_error(ctx)
    yield lib, ctx
    lib.futhark_context_free(ctx)
    lib.futhark_context_config_free(cfg)


def _i32_1d(lib, ctx, seq):
    a = np.ascontiguousarray(seq, dtype=np.int32)
    return lib.futhark_new_i32_1d(ctx, a.ctypes.data, a.size)


@pytest.mark.parametrize("case", OPS["query_sel"], ids=lambda c: c["id"])
def test_query_sel_through_futhark_names(fut, case):
    lib, ctx = fut
    db = np.ascontiguousarray(resolve_table(case["table"]), dtype=np.int32)
    arr = lib.futhark_new_i32_2d(ctx, db.ctypes.data, db.shape[0], db.shape[1])
    cols = _i32_1d(lib, ctx, case["cols"])
    out = C.c_void_p()
    assert lib.futhark_entry_query_sel(ctx, C.byref(out), arr, cols) == 0
    assert lib.futhark_context_sync(ctx) == 0
    shape = lib.futhark_shape_i32_2d(ctx, out)
    res = np.empty((shape[0], shape[1]), dtype=np.int32)
    assert lib.futhark_values_i32_2d(ctx, out, res.ctypes.data) == 0
    assert res.tolist() == case["out"]
    for h in (out, arr):
        lib.futhark_free_i32_2d(ctx, h)
    lib.futhark_free_i32_1d(ctx, cols)


@pytest.mark.parametrize("case", OPS["query_groupby"], ids=lambda c: c["id"])
def test_query_groupby_through_futhark_names(fut, case):
    lib, ctx = fut
    db = np.ascontiguousarray(resolve_table(case["table"]).astype(np.uint32))
    if db.size == 0:
        db = db.reshape(0, 2)
    arr = lib.futhark_new_u32_2d(ctx, db.ctypes.data, db.shape[0], db.shape[1])
    s, t = _i32_1d(lib, ctx, case["s_cols"]), _i32_1d(lib, ctx, case["t_cols"])
    out = C.c_void_p()
    assert lib.futhark_entry_query_groupby(ctx, C.byref(out), arr, int(case["g_col"]), s, t) == 0
    shape = lib.futhark_shape_u32_2d(ctx, out)
    res = np.empty((shape[0], shape[1]), dtype=np.uint32)
    assert lib.futhark_values_u32_2d(ctx, out, res.ctypes.data) == 0
    assert res.tolist() == case["out"]
    lib.futhark_free_u32_2d(ctx, out)
    lib.futhark_free_u32_2d(ctx, arr)
    lib.futhark_free_i32_1d(ctx, s)
    lib.futhark_free_i32_1d(ctx, t)


def test_join_through_futhark_names(fut, oracle):
    lib, ctx = fut
    rng = np.random.default_rng(5)
    a = rng.integers(0, 50, size=(300, 3)).astype(np.uint32)
    b = rng.integers(0, 50, size=(200, 2)).astype(np.uint32)
    A = lib.futhark_new_u32_2d(ctx, a.ctypes.data, *a.shape)
    B = lib.futhark_new_u32_2d(ctx, b.ctypes.data, *b.shape)
    c1, c2 = _i32_1d(lib, ctx, [0, 2]), _i32_1d(lib, ctx, [1])
    out = C.c_void_p()
    assert lib.futhark_entry_join(ctx, C.byref(out), A, B, 1, 0, c1, c2) == 0
    shape = lib.futhark_shape_u32_2d(ctx, out)
    res = np.empty((shape[0], shape[1]), dtype=np.uint32)
    assert lib.futhark_values_u32_2d(ctx, out, res.ctypes.data) == 0
    assert np.array_equal(res, oracle.join(a.astype(np.int64), b.astype(np.int64), 1, 0, [0, 2], [1]))
    # an entry's output is a valid input of the next entry (opaque arrays compose, as in Futhark)
    out2 = C.c_void_p()
    k = _i32_1d(lib, ctx, [1])
    t = _i32_1d(lib, ctx, [2])
    assert lib.futhark_entry_query_groupby(ctx, C.byref(out2), out, 0, k, t) == 0
    shape = lib.futhark_shape_u32_2d(ctx, out2)
    res2 = np.empty((shape[0], shape[1]), dtype=np.uint32)
    assert lib.futhark_values_u32_2d(ctx, out2, res2.ctypes.data) == 0
    assert np.array_equal(res2, oracle.query_groupby(res.astype(np.int64), 0, [1], [2]))
    for h in (out, out2, A, B):
        lib.futhark_free_u32_2d(ctx, h)


def test_error_string_is_callers_to_free(fut):
    lib, ctx = fut
    db = np.zeros((3, 2), dtype=np.int32)
    arr = lib.futhark_new_i32_2d(ctx, db.ctypes.data, 3, 2)
    cols = _i32_1d(lib, ctx, [5])
    out = C.c_void_p()
    assert lib.futhark_entry_query_sel(ctx, C.byref(out), arr, cols) != 0
    msg = lib.futhark_context_get_error(ctx)
    assert msg and b"col" in C.string_at(msg).lower() or msg
    C.CDLL(None).free(C.c_void_p(msg))
    lib.futhark_free_i32_2d(ctx, arr)
