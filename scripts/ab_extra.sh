#!/bin/bash
# A/B of a run-time switch on one box, headline AND kernel-fused entry point: scripts/ab_extra.sh OUTDIR VAR v1 v2 ...
# One JSON line per value in OUTDIR/abx_VAR.jsonl (headline evals/s, its passes, the fused entry point's evals/s and passes).
out=$1; var=$2; shift 2
mkdir -p "$out"
: > "$out/abx_$var.jsonl"
for v in "$@"; do
  env "$var=$v" timeout -k 10 240 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-ceiling > "$out/abx_tmp.json" 2> "$out/abx_tmp.err" || { echo "$var=$v failed"; tail -5 "$out/abx_tmp.err"; exit 1; }
  python - "$var" "$v" "$out/abx_tmp.json" >> "$out/abx_$var.jsonl" <<'PY'
import json, sys
r = json.load(open(sys.argv[3]))
k = {x["name"].split()[0]: round(1e3 * x["avg_ms_per_eval"], 1) for x in r["roofline"].get("kernels", [])}
f = r.get("extra", {}).get("kernel_fused_lpdf", {})
print(json.dumps({sys.argv[1]: sys.argv[2], "evals_per_s": round(r["value"], 1), "frac": round(r["roofline"]["frac"], 4), "us": k,
                  "fused_evals_per_s": round(f.get("evals_per_s", 0), 1),
                  "fused_us": {a: round(b, 1) for a, b in (f.get("pass_us_per_eval") or {}).items()}, "checked": r.get("checked")}))
PY
  tail -1 "$out/abx_$var.jsonl"
done
