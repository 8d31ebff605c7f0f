#!/bin/bash
# round 6, closing call: PMC passes of the implicit-GEMM family one shape per profiled process (the two convolution shapes launch the same grid size),
# the default bench line (with other_configs; quotes the PMC traffic of this tree), the UNETR++ published-block line, the full GPU suite + smoke
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06; mkdir -p $O
PM="FETCH_SIZE|WRITE_SIZE|SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE|SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
for shape in conv128 conv1024 linear; do
  IFS='|'; for c in $PM; do
    n=$(echo $c | tr ' ' '_'); unset IFS
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$n -- python3 tools/diagnostics/gemm_pmc.py $shape > $O/pmcg_${shape}_$n.log 2>&1
    IFS='|'
  done; unset IFS
  case $shape in
    conv128)  ALG='{"grid": 17072128}'; CM="conv3x3 128->128 on 2x128x128 (UNETR++ stage 0): input map 8.39 MB + output 8.39 MB + weight image 0.29 MB";;
    conv1024) ALG='{"grid": 27787264}'; CM="conv3x3 1024->1024 on 1x16x16, split-K (the kernel writes fp32 slabs, the epilogue is gemm_nt_reduce's): input 0.52 MB + weight image 18.9 MB + slabs 8.39 MB";;
    linear)   ALG='{"grid": 16809984}'; CM="Linear 32768x128 -> 128: every A row is read by exactly one tile through buffer_load ... lds (8.39 MB) + 8.39 MB out + 32 KB of weights: a known-size copy through the direct-to-LDS instruction -- ratio_fetch_as_counted ~ 0.76 and ratio_fetch_doubled ~ 1.02 mean the gfx950 FETCH_SIZE doubling DOES apply to it";;
  esac
  python3 tools/diagnostics/pmc_summary.py $O gemm_nt_kernel $O/pmc_traffic_gemm_nt_$shape.json $O/pmc_gemm_nt_$shape "$ALG" "gemm_pmc.py $shape: $CM" > $O/pmc_summary_gemm_nt_$shape.log 2>&1
  if [ $shape != linear ]; then
    python3 tools/diagnostics/pmc_summary.py $O gemm_tn_kernel $O/pmc_traffic_gemm_tn_$shape.json $O/pmc_gemm_tn_$shape '{"grid": 16777216}' "gemm_pmc.py $shape: weight gradient of the same convolution (x + dy: 16.8 MB at conv128, 1.05 MB at conv1024 -- the ratio of the latter is against the conv128 bytes and means nothing)" > $O/pmc_summary_gemm_tn_$shape.log 2>&1
  fi
  rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ_*
done
find $O -name "*.csv" -size +20M -delete
python3 bench.py > $O/halfunet_bf16_bench_default.json 2> $O/bench_default.err
python3 bench.py --model UNetRPP --strategy diff_ar --pred-steps 6 --steps 5 --warmup 2 --cpu-seconds 5 > $O/unetrpp_bf16_bench.json 2>/dev/null
python3 -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|^FAILED|^ERROR" | tail -12 > $O/gpu_tests.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke > $O/smoke.txt
cat $O/gpu_tests.txt $O/smoke.txt
python3 - <<'PY'
import json, glob
for f in ("halfunet_bf16_bench_default", "unetrpp_bf16_bench"):
    d = json.loads(open(f"gpurun_out/r06/{f}.json").readlines()[-1])
    print(f, round(d["value"], 2), round(d["ms_per_step"], 3), d["roofline"].get("frac") if d.get("roofline") else None, d["roofline"].get("traffic") if d.get("roofline") else None, d["config"].get("hip_graph_check"))
    if "other_configs" in d:
        print({k: (v.get("ms_per_step"), v.get("hip_graph")) for k, v in d["other_configs"].items() if isinstance(v, dict)})
for f in sorted(glob.glob("gpurun_out/r06/pmc_traffic_gemm_*_*.json")):
    d = json.load(open(f)); k = [x for x in d if x.startswith("gemm_")][0]
    print(f.split("/")[-1], {fl: (v["ratio_fetch_as_counted"], v["ratio_fetch_doubled"]) for fl, v in d[k].items()}, d.get("MfmaUtil_percent"))
PY
