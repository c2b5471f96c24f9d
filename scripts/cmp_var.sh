#!/bin/bash
# GPU box: time the headline bench with each gptools_amd/libvar_*.so (GPT_LIB), results may be wrong by construction
cd "$GRAFT_REPO_ROOT"
for f in gptools_amd/libgptools_hip.so gptools_amd/libvar_*.so; do
  [ -f "$f" ] || continue
  tag=$(basename $f .so)
  GPT_BENCH_NOCHECK=1 GPT_LIB=$PWD/$f timeout -k 10 200 python bench.py --no-cpu-baseline --steps 6 --warmup 2 "$@" > gpurun_out/var_$tag.json 2> gpurun_out/var_$tag.err || { echo "$tag failed"; tail -3 gpurun_out/var_$tag.err; exit 1; }
  python3 - "$tag" <<PY
import json,sys
r=json.load(open("gpurun_out/var_%s.json"%sys.argv[1]))
print(sys.argv[1], round(r["value"],1), " ".join("%s=%.1f"%(k["name"].split()[0], 1e3*k["avg_ms_per_eval"]) for k in r["roofline"].get("kernels", [])))
PY
done
