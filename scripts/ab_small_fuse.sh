#!/bin/bash
# A/B on one box: first / last stages of the small-lattice kernel reading / writing outside LDS (default) against the
# separate load / split / merge / store loops (GPT_SMALL_FUSE=0).
set -e
out=gpurun_out/ab_small_fuse
mkdir -p $out
for round in 1 2; do
  timeout -k 10 120 scripts/micro/bench_small_abi > $out/fused_$round.jsonl
  GPT_SMALL_FUSE=0 timeout -k 10 120 scripts/micro/bench_small_abi > $out/separate_$round.jsonl
done
python - <<'PY'
import json, glob
rows = {}
for f in sorted(glob.glob("gpurun_out/ab_small_fuse/*_[12].jsonl")):
    tag = f.split("/")[-1].rsplit("_", 1)[0]
    for line in open(f):
        d = json.loads(line)
        for k in ("us_per_call_device_ptr", "us_per_call_host_ptr_col_major"):
            if d[k] > 0:
                rows.setdefault(d["case"], {}).setdefault(tag + ":" + k, []).append(d[k])
with open("gpurun_out/ab_small_fuse/summary.jsonl", "w") as o:
    for case, t in rows.items():
        rec = {"case": case, **{k: min(v) for k, v in sorted(t.items())}}
        print(rec)
        o.write(json.dumps(rec) + "\n")
PY
