"""Print per-dispatch durations from a rocprofv3 kernel_trace.csv in launch order (grouped runs)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
import re
short = lambda n: (re.search(r"(fgb_\w+|digit_\w+|scan_\w+|filter_\w+|copy_\w+|\w+_kernel)", n) or re.search(r"\w+", n)).group(0)[:28]
prev = None; acc = []
for r in rows:
    name = short(r["Kernel_Name"]); d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if name.startswith("__amd") or name.startswith("gen_"): continue
    print(f"{name:30s} {d:10.1f} us  grid={r.get('Grid_Size_X','?')} wg={r.get('Workgroup_Size_X','?')} lds={r.get('LDS_Block_Size','?')} vgpr={r.get('VGPR_Count','?')}")
