"""Kernel / copy sequence after the LAST fgb_agg6 consumer launch in a rocprofv3 kernel_trace.csv (+ memory copies if traced): the tail of a statement.
Usage: python tools/tail_trace.py <dir-or-csv> [n_before]"""
import csv, glob, re, sys
p = sys.argv[1]
f = p if p.endswith(".csv") else glob.glob(p + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "fgb_part_kernel" in r["Kernel_Name"]]
j = idx[-1]
t0 = int(rows[j]["Start_Timestamp"])
for r in rows[j:]:
    nm = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); nm = re.sub(r"^void ", "", nm); nm = re.sub(r"\(.*", "", nm)[:50]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us +%8.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, nm))
