#!/bin/bash
# Power-of-two sizes beside the headline: cache-sized units (the rule for lattices of which >= 4 fit 128 MiB of workspace)
# against large units (GPT_UNIT_SMALL_MAX=100000 with GPT_UNIT_MIN_MIB=64: units of up to 2560 MiB), one box.
out=gpurun_out/ab_midsize; mkdir -p $out; : > $out/summary.jsonl
run() { # size batch label env...
  local n=$1 b=$2 label=$3; shift 3
  env "$@" timeout -k 10 200 python bench.py --size $n --batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-extra --no-ceiling > $out/tmp.json 2> $out/tmp.err || { echo "$label failed"; tail -3 $out/tmp.err; exit 1; }
  python - "$n" "$b" "$label" $out/tmp.json >> $out/summary.jsonl <<'PY'
import json, sys
r = json.load(open(sys.argv[4]))
k = {x["name"].split()[0]: round(1e3 * x["avg_ms_per_eval"], 2) for x in r["roofline"].get("kernels", [])}
print(json.dumps({"size": int(sys.argv[1]), "batch": int(sys.argv[2]), "variant": sys.argv[3], "evals_per_s": round(r["value"], 1), "frac": round(r["roofline"]["frac"], 4), "us": k, "checked": r.get("checked")}))
PY
  tail -1 $out/summary.jsonl
}
for nb in "256 4096" "512 1024" "1024 256" "2048 64"; do
  set -- $nb
  run $1 $2 default GPT_DUMMY=0
  run $1 $2 large_units GPT_UNIT_SMALL_MAX=100000 GPT_UNIT_MIN_MIB=64
  run $1 $2 units_512MiB GPT_UNIT_SMALL_MAX=100000 GPT_UNIT_MIN_MIB=64 GPT_UNIT_MIB=512
done
