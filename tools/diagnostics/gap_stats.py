"""Idle time of the main stream between consecutive kernels of one training step, by the kernel that FOLLOWS the gap.
Input: the rocpd SQLite database of `rocprofv3 --kernel-trace` (ROCm 7.2; view `kernels`: name, start, end, queue_id / stream_id).
usage: python tools/diagnostics/gap_stats.py <results.db> [first_fraction=0.5]
Only the last part of the trace is analysed (steady state); the stream with the most kernels is taken as the main stream."""
import collections
import sqlite3
import sys

db = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
con = sqlite3.connect(db)
cols = [r[1] for r in con.execute("pragma table_info(kernels)").fetchall()]
stream_col = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else None)
q = f"select name, start, end{', ' + stream_col if stream_col else ''} from kernels order by start"
rows = con.execute(q).fetchall()
if stream_col:
    cnt = collections.Counter(r[3] for r in rows)
    main = cnt.most_common(1)[0][0]
    side = [r for r in rows if r[3] != main]
    rows = [r for r in rows if r[3] == main]
    print(f"streams: {dict(cnt)}; main = {main}")
rows = rows[int(len(rows) * frac):]
t0, t1 = rows[0][1], rows[-1][2]
busy = sum(r[2] - r[1] for r in rows)
print(f"{len(rows)} kernels over {(t1 - t0) / 1e6:.2f} ms: busy {busy / 1e6:.2f} ms, idle {(t1 - t0 - busy) / 1e6:.2f} ms")
gaps = collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    gaps[b[0][:70]].append(max(0, b[1] - a[2]))
print("idle before kernel (total us, count, median us):")
for name, g in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:25]:
    g.sort()
    print(f"  {sum(g) / 1e3:9.1f} {len(g):6d} {g[len(g) // 2] / 1e3:7.2f}  {name}")
