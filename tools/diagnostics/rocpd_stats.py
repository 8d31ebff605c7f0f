"""rocprofv3 (ROCm 7.2) writes a rocpd SQLite database with --kernel-trace --stats; this exports its `top_kernels` view as the
kernel_stats CSV kept under profiles/ (Name, Calls, TotalDurationUs, AverageUs, Percentage).
usage: python tools/diagnostics/rocpd_stats.py <results.db> <out.csv> [<one_step_trace.csv>]
With the third argument, also writes every kernel dispatch of the LAST complete optimizer step (between the last two `adamw_kernel`
dispatches) in start order with its duration: the per-launch view in which the full-resolution launches of the roofline kernel can
be told from the coarse-level ones that share its name in the aggregate table."""
import csv
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
con = sqlite3.connect(db)
rows = con.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
    for r in rows:
        w.writerow([r[0], r[1], int(r[2]), round(r[3], 1), round(r[4], 3)])
print(f"{len(rows)} kernels -> {out}")

if len(sys.argv) > 3:
    ks = con.execute("select name, start, end from kernels order by start").fetchall()
    marks = [i for i, k in enumerate(ks) if "adamw_kernel" in k[0]]
    if len(marks) >= 2:
        lo, hi = marks[-2] + 1, marks[-1] + 1
        with open(sys.argv[3], "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Index", "Name", "StartUsFromStepBegin", "DurationUs"])
            t0 = ks[lo][1]
            for i, (name, st, en) in enumerate(ks[lo:hi]):
                w.writerow([i, name[:120], round((st - t0) / 1e3, 2), round((en - st) / 1e3, 2)])
        print(f"{hi - lo} dispatches of the last step -> {sys.argv[3]}")
