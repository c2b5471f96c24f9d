#!/bin/bash
# A/B of run-time switches on one box: scripts/ab_env.sh OUTDIR VAR v1 v2 ... -- runs bench.py once per value
# (no CPU baseline, no extras), one JSON line per value in OUTDIR/ab_VAR.jsonl. Stops at the first failure.
out=$1; var=$2; shift 2
mkdir -p "$out"
: > "$out/ab_$var.jsonl"
for v in "$@"; do
  env "$var=$v" timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra > "$out/ab_tmp.json" 2> "$out/ab_tmp.err" || { echo "$var=$v failed"; tail -5 "$out/ab_tmp.err"; exit 1; }
  python - "$var" "$v" "$out/ab_tmp.json" >> "$out/ab_$var.jsonl" <<'PY'
import json, sys
r = json.load(open(sys.argv[3]))
k = {x["name"].split()[0]: round(1e3 * x["avg_ms_per_eval"], 1) for x in r["roofline"].get("kernels", [])}
print(json.dumps({sys.argv[1]: sys.argv[2], "evals_per_s": round(r["value"], 1), "frac": round(r["roofline"]["frac"], 4), "us": k, "checked": r.get("checked")}))
PY
  tail -1 "$out/ab_$var.jsonl"
done
