#!/bin/bash
# Workgroup size of the small-lattice kernel (GPT_SMALL_THREADS) against microseconds per call at the C ABI.
set -e
out=gpurun_out/ab_small_threads
mkdir -p $out
for t in 0 64 128 192 256 320 384 448 512 576 640 768 1024; do
  GPT_SMALL_THREADS=$t timeout -k 10 120 scripts/micro/bench_small_abi > $out/t$t.jsonl
done
python - <<'PY'
import json, glob, re
rows = {}
for f in glob.glob("gpurun_out/ab_small_threads/t*.jsonl"):
    t = int(re.search(r"t(\d+)\.jsonl", f).group(1))
    for line in open(f):
        d = json.loads(line)
        rows.setdefault(d["case"], {})[t] = d["us_per_call_device_ptr"]
with open("gpurun_out/ab_small_threads/summary.jsonl", "w") as o:
    for case, t in rows.items():
        rec = {"case": case, "us_by_threads (0 = the library's rule)": dict(sorted(t.items()))}
        print(rec)
        o.write(json.dumps(rec) + "\n")
PY
