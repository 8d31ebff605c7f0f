"""Merge the per-workload TunableOp result files of tools/diagnostics/tune_gemms.sh into the file the package loads
(py4cast_amd/tuning/tunableop_gfx950.csv): validator lines must agree, one line per (operator, shape) -- the fastest if a shape was
tuned in several workloads; shapes for which the library's default won are dropped (they need no entry)."""
import glob, sys
out = sys.argv[1] if len(sys.argv) > 1 else "py4cast_amd/tuning/tunableop_gfx950.csv"
validators, best = None, {}
# (the GNN workloads were tuned too and ran 2 % SLOWER with their selections in place -- 44.5 -> 45.5 ms for HiLAM: not merged)
for f in sorted(glob.glob("gpurun_out/tunable/swinunetr_0.csv") + glob.glob("gpurun_out/tunable/unetrpp_0.csv")):
    lines = [l.strip() for l in open(f) if l.strip()]
    v = [l for l in lines if l.startswith("Validator,")]
    if validators is None:
        validators = v
    assert v == validators, f"{f}: validators differ"
    for l in lines:
        if l.startswith("Validator,"):
            continue
        op, shape, sol, t = l.split(",")
        if (op, shape) not in best or float(t) < float(best[(op, shape)][1]):
            best[(op, shape)] = (sol, t)
kept = {k: v for k, v in best.items() if v[0] != "Default"}
with open(out, "w") as fh:
    fh.write("\n".join(validators) + "\n")
    for (op, shape), (sol, t) in sorted(kept.items()):
        fh.write(f"{op},{shape},{sol},{t}\n")
print(f"{len(best)} shapes tuned, {len(kept)} with a non-default solution -> {out}")
