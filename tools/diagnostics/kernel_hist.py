"""Per-launch durations of the kernels whose name contains a substring, grouped by grid size, from a rocprofv3 rocpd database.
usage: python tools/diagnostics/kernel_hist.py <results.db> <substring>"""
import collections
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, grid_x, grid_y, grid_z, workgroup_x, duration from kernels where name like ?", ("%" + sys.argv[2] + "%",)).fetchall()
by = collections.defaultdict(list)
for n, gx, gy, gz, wx, d in rows:
    by[(n[:70], gx // max(wx, 1), gy, gz)].append(d / 1e3)
tot = sum(sum(v) for v in by.values())
print(f"{len(rows)} launches, {tot / 1e3:.2f} ms")
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print(f"{sum(v) / tot * 100:5.1f}%  n={len(v):5d}  workgroups {k[1]:5d} x {k[2]} x {k[3]}  min/med/max {v[0]:7.1f} {v[len(v) // 2]:7.1f} {v[-1]:7.1f} us  {k[0]}")
