"""Sum the FETCH_SIZE / WRITE_SIZE counters (KiB) of every dispatch of a profiled bench run per optimizer step (round 4).
usage: python tools/diagnostics/pmc_step_sum.py <dir with pmc_FETCH_SIZE/ and pmc_WRITE_SIZE/> <out.json>
Steps are delimited by the adamw_kernel dispatches; the steps before the first and after the last AdamW launch (initialisation, the
un-timed tagged step) are dropped, the rest averaged.  FETCH_SIZE is doubled (gfx950: 32-B requests of 16-B/lane streams are counted in
64-B units, MI355X_MICROARCH.md HBM section) -- the same correction as pmc_summary.py."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root, out = sys.argv[1:3]
res = {}
fam_tot = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(os.path.join(root, "pmc_" + c, "**", "*counter_collection.csv"), recursive=True)
    rows = []
    for f in files:
        rows += [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == c]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    steps, cur = [], defaultdict(float)
    for r in rows:
        name = r["Kernel_Name"]
        cur[name.split("(")[0][-60:]] += float(r["Counter_Value"])
        if "adamw_kernel" in name:
            steps.append(cur)
            cur = defaultdict(float)
    steps = steps[1:]          # the first delimited span holds the initialisation
    n = len(steps)
    tot = [sum(s.values()) for s in steps]
    fam = defaultdict(float)
    for s in steps:
        for k, v in s.items():
            fam[k] += v / max(n, 1)
    fam_tot[c] = dict(sorted(fam.items(), key=lambda kv: -kv[1])[:25])
    res[c] = {"steps": n, "KiB_per_step": tot, "mean_KiB_per_step": sum(tot) / max(n, 1)}
if "FETCH_SIZE" in res and "WRITE_SIZE" in res:
    fetch = 2.0 * res["FETCH_SIZE"]["mean_KiB_per_step"] * 1024
    write = res["WRITE_SIZE"]["mean_KiB_per_step"] * 1024
    res["hbm_bytes_per_step"] = {"read": round(fetch), "written": round(write), "total": round(fetch + write)}
res["largest_kernels_KiB_per_step"] = fam_tot
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from py4cast_amd._lib import kernel_sources_sha16   # noqa: E402

res["kernel_sources_sha16"] = kernel_sources_sha16()   # bench.py reports the figure only for a tree with these kernel sources
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "largest_kernels_KiB_per_step"}, indent=1))
