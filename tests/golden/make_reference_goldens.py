"""Generate tests/golden/reference_host.json by RUNNING the reference's own host code.

Run in the build container only (the reference tree is not on the GPU box):
    python tests/golden/make_reference_goldens.py /root/reference

What runs: the reference's parse.py (sql_parse) and table.py (Table / load_*), imported unmodified from the
reference checkout.  parse.py imports `moz_sql_parser`, which is not installed here: the module is stubbed with
harkdb_amd.sqlfront.parse, which emits the parse-tree shape parse.py indexes (parse.py:29, :46-51, :66, :72-84).
So the goldens pin "the reference planner's output for these parse trees", not moz_sql_parser itself.
Only data is written: statements, inputs, and the IR / schema / values / error text the reference produced.
"""
import io, json, os, sys, types, contextlib
import numpy as np
import pandas as pd

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(here)))
from harkdb_amd import sqlfront

stub = types.ModuleType("moz_sql_parser")
stub.parse = sqlfront.parse
sys.modules["moz_sql_parser"] = stub
sys.path.insert(0, ref)
import parse as ref_parse          # noqa: E402  (the reference's parse.py)
import table as ref_table          # noqa: E402  (the reference's table.py)

csv = os.path.join(here, "data.csv")            # byte-identical copy of the reference's data.csv
tables = {"game_1": ref_table.Table("game_1", csv)}

STATEMENTS = [
    "select col1, col3 from game_1",
    "select col3, col1, col8 from game_1",
    "select col1,  max(col3) from game_1 group by col1",
    "select col1, sum(col3), min(col2), prod(col7) from game_1 group by col1",
    "select prod(col2), sum(col3), min(col4) from game_1 group by col8",
    "select max(col1), col2 from game_1 group by col2",
    "select col1, col3 from game_1 where col2 > 3",
    "select col1, col3 from game_1 order by col1",
    "select col1, count(col3) from game_1 group by col1",
    "select col1 from game_1",
    "select * from game_1",
    "select col1 from nope",
    "select colx, col1 from game_1",
    "select max(col1) from game_1 group by colx",
    "select col1, col2 from game_1 group by col1",
    "select col1, max(colx) from game_1 group by col1",
]


def describe(ir):
    out = {}
    for k, v in ir.items():
        if k == "table":
            a = np.asarray(v)
            out[k] = {"shape": list(a.shape), "dtype": str(a.dtype), "sum": int(a.sum())}
        else:
            out[k] = np.asarray(v).tolist() if isinstance(v, (list, np.ndarray)) else v
    return out


planner = []
for stmt in STATEMENTS:
    buf = io.StringIO()
    try:
        with contextlib.redirect_stdout(buf):
            ir = ref_parse.sql_parse(tables, stmt)
        planner.append({"sql": stmt, "ir": describe(ir)})
    except Exception as e:                                   # noqa: BLE001 -- the reference raises plain Exception / TypeError
        planner.append({"sql": stmt, "raises": type(e).__name__, "message": str(e)})

ingest = []
df = pd.DataFrame({"a": [1, 2, 3], "b": [4, 5, 6]})
arr = np.arange(12).reshape(3, 4)
for name, obj in (("dataframe", df), ("ndarray", arr), ("csv", csv)):
    t = ref_table.Table("t", obj)
    d = np.asarray(t.get_data())
    ingest.append({"input": name, "schema": list(t.get_schema()), "shape": list(d.shape), "dtype": str(d.dtype),
                   "values": d.tolist(), "name": t.get_name()})
try:
    ref_table.Table("t", 3.5)
except Exception as e:                                       # noqa: BLE001
    ingest.append({"input": "float", "raises": type(e).__name__, "message": str(e)})
try:
    ref_table.Table("t", "x.parquet")
except Exception as e:                                       # noqa: BLE001
    ingest.append({"input": "x.parquet", "raises": type(e).__name__, "message": str(e)})

json.dump({"_source": "Output of the reference's parse.py / table.py run by tests/golden/make_reference_goldens.py "
                      "(moz_sql_parser stubbed with harkdb_amd.sqlfront.parse). Data only.",
           "planner": planner, "ingest": ingest}, open(os.path.join(here, "reference_host.json"), "w"), indent=1)
print("wrote", len(planner), "planner and", len(ingest), "ingest goldens")
