"""Record what the reference's FutharkContext hands to its native entries (FutharkContext.py:65-66, :70-71).

    python tests/golden/make_reference_calls.py /root/reference        (build container only)

The reference's FutharkContext.py, parse.py and table.py run unmodified.  `futhark_ffi.Futhark` and `_main` (the
generated native module, absent here) are replaced by a recorder that notes the entry name and the arguments of
every call; `moz_sql_parser` is stubbed with harkdb_amd.sqlfront.parse.  Output: tests/golden/reference_calls.json
(data only: statements and the recorded calls).
"""
import contextlib, io, json, os, sys, types
import numpy as np

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(here)))
from harkdb_amd import sqlfront

calls = []


def describe(a):
    if isinstance(a, np.ndarray):
        return {"ndarray": {"shape": list(a.shape), "dtype": str(a.dtype), "f_contiguous": bool(a.flags.f_contiguous and a.ndim > 1),
                            "values": a.tolist() if a.size <= 64 else None, "sum": int(a.sum())}}
    return {"python": type(a).__name__, "value": a}


class Recorder:
    def __init__(self, module):
        pass

    def __getattr__(self, entry):
        def call(*args):
            calls.append({"entry": entry, "args": [describe(a) for a in args]})
            return ("result-of", entry)
        return call

    def from_futhark(self, res):
        calls.append({"entry": "from_futhark", "args": [list(res)]})
        return res


for name, attrs in (("moz_sql_parser", {"parse": sqlfront.parse}), ("futhark_ffi", {"Futhark": Recorder}), ("_main", {})):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
sys.path.insert(0, ref)
import FutharkContext as ref_ctx        # noqa: E402  (the reference's FutharkContext.py)

fc = ref_ctx.FutharkContext()
fc.create_table("game_1", os.path.join(here, "data.csv"))
out = []
for stmt in ("select col1, col3 from game_1",                           # README.md:42
             "select col1,  max(col3) from game_1 group by col1",        # test.py:7
             "select col1, sum(col3), min(col2), prod(col7) from game_1 group by col1"):
    calls.clear()
    with contextlib.redirect_stdout(io.StringIO()):
        fc.sql(stmt)
    out.append({"sql": stmt, "calls": json.loads(json.dumps(calls))})
fc.drop_table("game_1")
out.append({"tables_after_drop": sorted(fc.tables)})
json.dump({"_source": "Calls made by the reference's FutharkContext.sql(), recorded by tests/golden/make_reference_calls.py "
                      "(native layer and SQL parser stubbed). Data only.", "statements": out},
          open(os.path.join(here, "reference_calls.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:1800])
