"""rocprofv3 (ROCm 7.2) writes a rocpd SQLite database with --kernel-trace --stats; this exports its `top_kernels` view as the
kernel_stats CSV kept under profiles/ (Name, Calls, TotalDurationNs, AverageNs, Percentage).
usage: python tools/diagnostics/rocpd_stats.py <results.db> <out.csv>"""
import csv
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
con = sqlite3.connect(db)
rows = con.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
    for r in rows:
        w.writerow([r[0], r[1], int(r[2]), round(r[3], 1), round(r[4], 3)])
print(f"{len(rows)} kernels -> {out}")
