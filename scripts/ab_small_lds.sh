#!/bin/bash
# A/B on one box: small-lattice path with LDS-typed line buffers, inlined line transforms and twiddle tables in LDS
# (this tree) against the library of the previous commit (gptools_amd/old/, built from a worktree of that commit).
#   new        this tree

#   old        previous commit
set -e
out=gpurun_out/ab_small_lds
mkdir -p $out
for round in 1 2; do
  timeout -k 10 120 scripts/micro/bench_small_abi > $out/new_$round.jsonl

  LD_LIBRARY_PATH=gptools_amd/old timeout -k 10 120 scripts/micro/bench_small_abi > $out/old_$round.jsonl
done
python - <<'PY'
import json, glob
rows = {}
for f in sorted(glob.glob("gpurun_out/ab_small_lds/*.jsonl")):
    tag = f.split("/")[-1].rsplit("_", 1)[0]
    for line in open(f):
        d = json.loads(line)
        rows.setdefault(d["case"], {}).setdefault(tag, []).append(d["us_per_call_device_ptr"])
with open("gpurun_out/ab_small_lds/summary.jsonl", "w") as o:
    for case, t in rows.items():
        rec = {"case": case, **{k: min(v) for k, v in t.items()}}
        print(rec)
        o.write(json.dumps(rec) + "\n")
PY
