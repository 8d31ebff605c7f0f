"""Per-launch-site averages from a rocprofv3 kernel trace: consecutive dispatches of one kernel with one grid are one 'run'.
Usage: python tools/diagnostics/trace_runs.py <kernel_trace.csv> [name filter] [min run length]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
minlen = int(sys.argv[3]) if len(sys.argv) > 3 else 5
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seq = []
for r in rows:
    n = r["Kernel_Name"].replace("void p4c::", "").replace("(anonymous namespace)::", "")[:70]
    g = (int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), int(r["Grid_Size_Y"]))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if seq and seq[-1][0] == (n, g):
        seq[-1][1].append(d)
    else:
        seq.append([(n, g), [d]])
for (n, g), ds in seq:
    if flt in n and len(ds) >= minlen:
        tail = ds[len(ds) // 5:]
        print(f"{n:70s} wgs {g[0]:6d} x {g[1]:3d}  n={len(ds):4d}  avg {sum(tail) / len(tail):8.2f} us  min {min(ds):8.2f}")
