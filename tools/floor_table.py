"""Per-kernel table: bytes moved (PMC) against the plain-stream floor of those bytes on this card -- reads at 6.9 TB/s (the
three-column read probe of bench.py), writes beside reads at 2.8 TB/s (the marginal cost of the written bytes in the
read + write probe: profiles/r04_notes.md 12) -- from profiles/<tag>_op_traffic.txt (tools/opmc.py output).
Usage: python tools/floor_table.py profiles/r04_op_traffic.txt > profiles/r04_floor_table.md"""
import re, sys
READ_TBS, WRITE_TBS = 6.9, 2.8
print("Kernels of the operator workloads (>= 40 us) against what plain streams need for the same bytes on this card: reads at 6.9 TB/s,\nwrites beside reads at 2.8 TB/s (`bench.py` probes; a pure write stream reaches 4.5, so short kernels can come in under the floor).\n")
print("| workload | kernel | calls | read MB | written MB | measured us | floor us (read / 6.9 + written / 2.8 TB/s) | measured / floor |")
print("|---|---|---|---|---|---|---|---|")
work = None
for line in open(sys.argv[1]):
    if line.startswith("== "):
        work = line[3:].strip(); continue
    m = re.match(r"^(\S.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", line)
    if not m or work is None:
        continue
    name, calls, rd, wr, us = m.group(1), int(m.group(2)), float(m.group(3)), float(m.group(4)), float(m.group(5))
    if us < 40.0:
        continue
    floor = rd / READ_TBS + wr / WRITE_TBS            # MB / (TB/s) = us
    print(f"| {work} | `{name}` | {calls} | {rd:.0f} | {wr:.0f} | {us:.0f} | {floor:.0f} | {us / floor:.2f} |")
