"""Summarise the separate `rocprofv3 --pmc <group> --kernel-trace --output-format csv` passes of tools/diagnostics/conv_exp.py into
the per-launch figures kept as profiles/rNN_pmc_traffic.json, and keep the raw counter rows of the roofline kernel.
usage: python tools/diagnostics/pmc_summary.py <dir with pmc_<GROUP>/ sub-directories> <kernel substring> <out.json> <raw rows dir>
       [algorithmic bytes per launch | JSON {"flavour substring": bytes, ...}] [comment for the file]
(round 6: per-flavour algorithmic bytes and the comment are arguments -- the round-5 gemm files carried the row convolution's)
FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE counts 32-B requests as 64-B units for 16-B/lane streaming reads and is doubled
(MI355X_MICROARCH.md, HBM section)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root, kern, out, rawdir = sys.argv[1:5]
os.makedirs(rawdir, exist_ok=True)
vals = defaultdict(lambda: defaultdict(list))   # counter -> template flavour -> values
for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if kern in r["Kernel_Name"]]
        if not rows:
            continue
        keep = ["Dispatch_Id", "Kernel_Name", "Grid_Size", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"]
        with open(os.path.join(rawdir, os.path.basename(d).replace("pmc_", kern + "_") + ".csv"), "w", newline="") as g:
            w = csv.DictWriter(g, keep)
            w.writeheader()
            for r in rows:
                w.writerow({k: (r[k][:100] if k == "Kernel_Name" else r[k]) for k in keep})
        for r in rows:
            flavour = r["Kernel_Name"].split(kern, 1)[1].split("(", 1)[0] + " grid " + str(int(float(r["Grid_Size"])) // 512) + " wg"
            vals[r["Counter_Name"]][flavour].append(float(r["Counter_Value"]))

def mean(v):
    return sum(v) / len(v) if v else None

comment = sys.argv[6] if len(sys.argv) > 6 else ("per-launch means over the launches of tools/diagnostics/conv_exp.py (3x3 conv 64->64, bf16 "
                                                 "activations, B=2, 512x512)")
res = {"_comment": comment + "; one rocprofv3 --pmc pass per counter group with --kernel-trace only; FETCH_SIZE/WRITE_SIZE in KiB, FETCH_SIZE "
                   "doubled on gfx950 where the reads are 16-byte-per-lane streams (MI355X_MICROARCH.md)",
       "per_flavour": {c: {fl: {"mean": mean(v), "launches": len(v)} for fl, v in d.items()} for c, d in vals.items()}}
allf = [x for v in vals.get("FETCH_SIZE", {}).values() for x in v]
allw = [x for v in vals.get("WRITE_SIZE", {}).values() for x in v]
per_flavour_alg = None
if len(sys.argv) > 5 and sys.argv[5].lstrip().startswith("{"):
    per_flavour_alg = json.loads(sys.argv[5])
if allf and allw and per_flavour_alg is None:
    hbm = (2.0 * mean(allf) + mean(allw)) * 1024
    alg = float(sys.argv[5]) if len(sys.argv) > 5 else 2.0 * 64 * 2 * 2 * 512 * 512
    res[kern] = {"FETCH_SIZE_KiB": mean(allf), "WRITE_SIZE_KiB": mean(allw), "hbm_bytes_per_launch": round(hbm),
                 "algorithmic_bytes_per_launch": alg, "ratio": round(hbm / alg, 3)}
if per_flavour_alg is not None:
    res[kern] = {}
    for fl in vals.get("FETCH_SIZE", {}):
        f_, w_ = mean(vals["FETCH_SIZE"][fl]), mean(vals.get("WRITE_SIZE", {}).get(fl, []))
        alg = next((b for k, b in per_flavour_alg.items() if k in fl), None)
        if w_ is None or alg is None:
            continue
        res[kern][fl] = {"FETCH_SIZE_KiB": f_, "WRITE_SIZE_KiB": w_, "algorithmic_bytes_per_launch": alg,
                         "hbm_bytes_per_launch_fetch_as_counted": round((f_ + w_) * 1024), "ratio_fetch_as_counted": round((f_ + w_) * 1024 / alg, 3),
                         "hbm_bytes_per_launch_fetch_doubled": round((2 * f_ + w_) * 1024), "ratio_fetch_doubled": round((2 * f_ + w_) * 1024 / alg, 3)}
busy, act = vals.get("SQ_VALU_MFMA_BUSY_CYCLES", {}), vals.get("GRBM_GUI_ACTIVE", {})
if busy and act:
    res["MfmaUtil_percent"] = {fl: round(100.0 * (mean(busy[fl]) / 1024) / (mean(act[fl]) / 8), 1) for fl in busy if fl in act}
    res["MfmaUtil_note"] = "SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs over GRBM_GUI_ACTIVE / 8 XCDs"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from py4cast_amd._lib import kernel_sources_sha16   # noqa: E402

res["kernel_sources_sha16"] = kernel_sources_sha16()   # bench.py reports the figure only for a tree with these kernel sources
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1)[:3000])
