#!/bin/bash
# Column-kernel geometry (GPT_SUB_GEOM) against the number of lattices per launch at 4096^2: profiles/r03_j_batch_geom_ab.jsonl
out=gpurun_out/r03j; mkdir -p $out; : > $out/batch_geom.jsonl
for b in 1 4; do for g in 0 1 3 4; do
  GPT_SUB_GEOM=$g timeout -k 10 200 python bench.py --batch $b --steps 20 --warmup 3 --no-cpu-baseline --no-extra > $out/tmp.json 2> $out/tmp.err || { echo fail; tail -3 $out/tmp.err; exit 1; }
  python - $b $g $out/tmp.json >> $out/batch_geom.jsonl <<'PY'
import json, sys
r = json.load(open(sys.argv[3]))
k = {x["name"].split()[0]: round(1e3 * x["avg_ms_per_eval"], 1) for x in r["roofline"].get("kernels", [])}
print(json.dumps({"batch": int(sys.argv[1]), "GPT_SUB_GEOM": int(sys.argv[2]), "evals_per_s": round(r["value"], 1), "us": k}))
PY
  tail -1 $out/batch_geom.jsonl
done; done
