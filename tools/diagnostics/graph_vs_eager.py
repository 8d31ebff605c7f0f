"""Compare two per-dispatch timelines of ONE optimizer step (tools/diagnostics/step_timeline.py's csv): eager launching vs HIP-graph replay.
usage: python graph_vs_eager.py <eager.csv> <graph.csv>"""
import csv, sys, collections, re

def load(p):
    rows = list(csv.DictReader(open(p)))
    for r in rows:
        r["s"], r["d"] = float(r["StartUs"]), float(r["DurationUs"])
    return rows

def short(n):
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*", "", n)[:60]

for tag, p in (("eager", sys.argv[1]), ("graph replay", sys.argv[2])):
    rows = load(p)
    byq = collections.defaultdict(list)
    for r in rows:
        byq[r["Queue"]].append(r)
    span = max(r["s"] + r["d"] for r in rows) - min(r["s"] for r in rows)
    print(f"== {tag}: {len(rows)} dispatches on {len(byq)} queue(s), span {span:.0f} us, sum of kernel durations {sum(r['d'] for r in rows):.0f} us")
    for q, v in sorted(byq.items(), key=lambda kv: -len(kv[1])):
        v.sort(key=lambda r: r["s"])
        gaps = [v[i + 1]["s"] - (v[i]["s"] + v[i]["d"]) for i in range(len(v) - 1)]
        pos = [g for g in gaps if g > 0]
        pos.sort()
        med = pos[len(pos) // 2] if pos else 0
        print(f"   queue {q}: {len(v)} kernels, busy {sum(r['d'] for r in v):.0f} us, gaps: total {sum(pos):.0f} us, median {med:.2f} us, "
              f"> 5 us: {sum(g > 5 for g in pos)}, > 20 us: {sum(g > 20 for g in pos)}")
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in rows:
        a = agg[short(r["Name"])]
        a[0] += 1; a[1] += r["d"]
    globals()["agg_" + tag.split()[0]] = agg
print("== per kernel: calls, average duration eager -> graph (us)")
for k, (c, t) in sorted(agg_eager.items(), key=lambda kv: -kv[1][1])[:22]:
    g = agg_graph.get(k, [0, 0.0])
    print(f"   {c:4d} {t / c:8.1f} -> {g[0]:4d} {(g[1] / g[0]) if g[0] else 0:8.1f}   {k}")
