#!/bin/bash
# GPU box: time the headline bench with each sub-line variant (GPT_SUBLINE=0/2/4, optional GPT_LIB builds)
cd "$GRAFT_REPO_ROOT"
run() {
  tag=$1; shift
  env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/cmp_$tag.json 2> gpurun_out/cmp_$tag.err || { echo "$tag failed"; tail -3 gpurun_out/cmp_$tag.err; return 1; }
  python3 - "$tag" <<PY
import json,sys
r=json.load(open("gpurun_out/cmp_%s.json"%sys.argv[1]))
print(sys.argv[1], round(r["value"],1), " ".join("%s=%.1f"%(k["name"].split()[0], 1e3*k["avg_ms_per_eval"]) for k in r["roofline"].get("kernels",[])), "frac=%.4f"%r["roofline"]["frac"])
PY
}
run whole GPT_SUBLINE=0 && run sub2 GPT_SUBLINE=2 && run sub4 GPT_SUBLINE=4
for f in gptools_amd/libvar_*.so; do [ -f "$f" ] && run $(basename $f .so) GPT_SUBLINE=2 GPT_LIB=$PWD/$f; done
