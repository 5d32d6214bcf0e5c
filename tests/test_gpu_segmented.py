"""The reference's 18 known-answer vectors (segmented_tests.fut:5-72) against the
HIP implementations of the vendored segmented primitives, plus larger random
inputs against the oracle.  Bit-exact."""
import ctypes as C

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu
KAT = load_golden("segmented_kat.json")


@pytest.fixture(scope="module")
def eng():
    from harkdb_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


class Dev:
    """Tiny helper: numpy array <-> device buffer."""

    def __init__(self, eng):
        self.eng, self.bufs = eng, []

    def up(self, arr):
        arr = np.ascontiguousarray(arr)
        ptr = self.eng.alloc(max(arr.nbytes, 16))
        if arr.nbytes:
            self.eng.upload(ptr, arr)
        self.bufs.append(ptr)
        return ptr

    def new(self, n, dtype):
        ptr = self.eng.alloc(max(n * np.dtype(dtype).itemsize, 16))
        self.bufs.append(ptr)
        return ptr

    def free(self):
        for b in self.bufs:
            self.eng.free(b)


def seg_scan(eng, flags, vals):
    d = Dev(eng)
    f, v = np.asarray(flags, np.uint8), np.asarray(vals, np.int32)
    out = d.new(v.size, np.int32)
    eng._chk(eng.lib.hark_op_segmented_scan_add_i32(eng.ctx, d.up(f), d.up(v), v.size, out))
    res = eng.download(out, v.size, np.int32)
    d.free()
    return res


def seg_reduce(eng, flags, vals):
    d = Dev(eng)
    f, v = np.asarray(flags, np.uint8), np.asarray(vals, np.int32)
    out, n = d.new(v.size, np.int32), C.c_int64()
    eng._chk(eng.lib.hark_op_segmented_reduce_add_i32(eng.ctx, d.up(f), d.up(v), v.size, out, C.byref(n)))
    res = eng.download(out, n.value, np.int32)
    d.free()
    return res


def repl_iota(eng, reps):
    d = Dev(eng)
    r = np.asarray(reps, np.int32)
    n = C.c_int64()
    rp = d.up(r)
    eng._chk(eng.lib.hark_op_replicated_iota(eng.ctx, rp, r.size, None, C.byref(n)))
    out = d.new(n.value, np.int32)
    eng._chk(eng.lib.hark_op_replicated_iota(eng.ctx, rp, r.size, out, C.byref(n)))
    res = eng.download(out, n.value, np.int32)
    d.free()
    return res


def seg_iota(eng, flags):
    d = Dev(eng)
    f = np.asarray(flags, np.uint8)
    out = d.new(f.size, np.int32)
    eng._chk(eng.lib.hark_op_segmented_iota(eng.ctx, d.up(f), f.size, out))
    res = eng.download(out, f.size, np.int32)
    d.free()
    return res


def expand(eng, arr):
    """test_expand of segmented_tests.fut:55-56: sz = id, get = x * i."""
    d = Dev(eng)
    a = np.asarray(arr, np.int32)
    n = C.c_int64()
    ap = d.up(a)
    eng._chk(eng.lib.hark_op_expand_indices(eng.ctx, ap, a.size, None, None, C.byref(n)))
    idxs, iotas = d.new(n.value, np.int32), d.new(n.value, np.int32)
    eng._chk(eng.lib.hark_op_expand_indices(eng.ctx, ap, a.size, idxs, iotas, C.byref(n)))
    i, j = eng.download(idxs, n.value, np.int32), eng.download(iotas, n.value, np.int32)
    d.free()
    return a[i] * j if n.value else np.empty(0, np.int32)


@pytest.mark.parametrize("case", KAT["segmented_scan"], ids=lambda c: c["ref"])
def test_kat_segmented_scan(eng, case):
    assert seg_scan(eng, case["flags"], case["as"]).tolist() == case["out"]


@pytest.mark.parametrize("case", KAT["segmented_reduce"], ids=lambda c: c["ref"])
def test_kat_segmented_reduce(eng, case):
    assert seg_reduce(eng, case["flags"], case["as"]).tolist() == case["out"]


@pytest.mark.parametrize("case", KAT["replicated_iota"], ids=lambda c: c["ref"])
def test_kat_replicated_iota(eng, case):
    assert repl_iota(eng, case["in"]).tolist() == case["out"]


@pytest.mark.parametrize("case", KAT["segmented_iota"], ids=lambda c: c["ref"])
def test_kat_segmented_iota(eng, case):
    assert seg_iota(eng, case["flags"]).tolist() == case["out"]


@pytest.mark.parametrize("case", KAT["expand"], ids=lambda c: c["ref"])
def test_kat_expand(eng, case):
    assert expand(eng, case["in"]).tolist() == case["out"]


@pytest.mark.parametrize("n,density", [(1, 1.0), (4095, 0.3), (4097, 0.001), (100_003, 0.05), (1_000_000, 0.0), (262_144, 1.0)])
def test_random_against_oracle(eng, oracle, n, density):
    rng = np.random.default_rng(n)
    flags = (rng.random(n) < density).astype(np.uint8)
    vals = rng.integers(-2**31, 2**31, size=n).astype(np.int32)          # wrap-around sums included
    assert np.array_equal(seg_scan(eng, flags, vals), oracle.segmented_scan_add(flags, vals))
    assert np.array_equal(seg_reduce(eng, flags, vals), oracle.segmented_reduce_add(flags, vals))
    assert np.array_equal(seg_iota(eng, flags), oracle.segmented_iota(flags))


@pytest.mark.parametrize("n", [1, 17, 5000])
def test_replicated_iota_and_expand_random(eng, oracle, n):
    rng = np.random.default_rng(n + 1)
    reps = rng.integers(0, 6, size=n).astype(np.int32)
    assert np.array_equal(repl_iota(eng, reps), oracle.replicated_iota(reps))
    assert np.array_equal(expand(eng, reps), oracle.test_expand(reps))
