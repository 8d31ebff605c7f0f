"""Critical-path view of ONE optimizer step from a rocprofv3 rocpd database: per stream (queue) busy time, gaps between
consecutive kernels of the main stream, time where only the side stream runs.
usage: python tools/diagnostics/step_timeline.py <results.db> [out.csv] [name of the step's last kernel]"""
import sqlite3, sys, collections, re

con = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in con.execute("pragma table_info(kernels)").fetchall()]
qcol = next((c for c in ("queue_id", "stream_id", "queue", "stream") if c in cols), None)
ks = con.execute(f"select name, start, end, {qcol or 0} from kernels order by start").fetchall()
marker = sys.argv[3] if len(sys.argv) > 3 else "adamw_kernel"   # the LAST kernel of a step (one launch per step)
marks = [i for i, k in enumerate(ks) if marker in k[0]]
if len(marks) < 2:
    cnt = collections.Counter(k[0] for k in ks)
    print("marker %r: %d launches; candidates (kernels launched a few times):" % (marker, len(marks)))
    for n, c in sorted(cnt.items(), key=lambda kv: kv[1])[:30]:
        print("   %5d  %s" % (c, n[:150]))
    sys.exit(1)
lo, hi = marks[-2] + 1, marks[-1] + 1
step = ks[lo:hi]
t0 = step[0][1]
byq = collections.defaultdict(list)
for n, s, e, q in step:
    byq[q].append((n, (s - t0) / 1e3, (e - t0) / 1e3))
main_q = max(byq, key=lambda q: len(byq[q]))
print("columns:", cols)
print("step: %d kernels, span %.0f us" % (len(step), (step[-1][2] - t0) / 1e3))
for q, v in byq.items():
    print("  queue %s: %d kernels, busy %.0f us%s" % (q, len(v), sum(e - s for _, s, e in v), "  (main)" if q == main_q else ""))
m = byq[main_q]
gaps = [(m[i + 1][1] - m[i][2], m[i][0], m[i + 1][0]) for i in range(len(m) - 1)]
print("main-stream gaps: total %.0f us, >5us: %d, >20us: %d" % (sum(max(0, g[0]) for g in gaps), sum(g[0] > 5 for g in gaps), sum(g[0] > 20 for g in gaps)))
def short(n):
    n = re.sub(r"^void ", "", n)
    m_ = re.match(r"_ZN3p4c(?:12_GLOBAL__N_1)?\d+([A-Za-z0-9_]+?)I", n)
    return (m_.group(1) if m_ else n)[:44]
print("largest gaps on the main stream (us, after -> before):")
for g in sorted(gaps, key=lambda g: -g[0])[:25]:
    print("   %7.1f  %-44s -> %s" % (g[0], short(g[1]), short(g[2])))
# intervals where main is idle: what runs on other queues
others = [x for q, v in byq.items() if q != main_q for x in v]
idle_cov = 0.0
for i in range(len(m) - 1):
    a, b = m[i][2], m[i + 1][1]
    if b - a <= 0:
        continue
    for _, s, e in others:
        idle_cov += max(0.0, min(b, e) - max(a, s))
print("main-stream idle time covered by side-stream kernels: %.0f us" % idle_cov)
if len(sys.argv) > 2:
    import csv
    with open(sys.argv[2], "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Index", "Queue", "Name", "StartUs", "DurationUs"])
        for i, (n, s, e, q) in enumerate(step):
            w.writerow([i, q, n[:100], round((s - t0) / 1e3, 2), round((e - s) / 1e3, 2)])
