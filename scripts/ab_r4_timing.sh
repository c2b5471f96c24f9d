#!/bin/bash
# RECORD ONLY: the knobs it sets (GPT_SUB_TIMING_R4, radix-4 sub-lines) existed in commit f8a08d8 alone; kept so that
# profiles/r03_g_r4_timing.jsonl can be traced to what produced it. Timing-only A/B (results of the R4 runs are garbage).
out=$1; mkdir -p "$out"; : > "$out/r4_timing.jsonl"
run() { # label, env...
  label=$1; shift
  env "$@" GPT_BENCH_NOCHECK=1 timeout -k 10 240 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-ceiling > "$out/r4_tmp.json" 2> "$out/r4_tmp.err" || { echo "$label failed"; tail -5 "$out/r4_tmp.err"; exit 1; }
  python - "$label" "$out/r4_tmp.json" >> "$out/r4_timing.jsonl" <<'PY'
import json, sys
r = json.load(open(sys.argv[2]))
k = {x["name"].split()[0]: round(1e3 * x["avg_ms_per_eval"], 1) for x in r["roofline"].get("kernels", [])}
f = r.get("extra", {}).get("kernel_fused_lpdf", {})
print(json.dumps({"run": sys.argv[1], "evals_per_s": round(r["value"], 1), "us": k, "fused_evals_per_s": round(f.get("evals_per_s", 0), 1),
                  "fused_us": {a: round(b, 1) for a, b in (f.get("pass_us_per_eval") or {}).items()}}))
PY
  tail -1 "$out/r4_timing.jsonl"
}
run "shipped" GPT_X=0
run "R4 timing, 256 threads" GPT_SUB_TIMING_R4=1 GPT_SUB_GEOM=1
run "R4 timing, 256 threads, nt" GPT_SUB_TIMING_R4=1 GPT_SUB_GEOM=2
run "R4 timing, 512 threads (16 columns), nt" GPT_SUB_TIMING_R4=1 GPT_SUB_GEOM=4
run "traffic only, shipped geometry" GPT_LIB=$PWD/gptools_amd/libvar_traffic.so
run "traffic only, 256 threads (4 columns)" GPT_LIB=$PWD/gptools_amd/libvar_traffic.so GPT_SUB_GEOM=1
