#!/bin/bash
# GPU box: headline bench with several unit sizes of the forward row pass (GPT_UNIT_MIB of workspace; 128 = one
# 4096^2 lattice).  usage: scripts/cmp_ws.sh "<bench args>" <MiB>...
cd "$GRAFT_REPO_ROOT"
args=$1; shift
for m in "$@"; do
  GPT_UNIT_MIB=$m timeout -k 10 200 python bench.py --no-cpu-baseline $args 2>/dev/null > gpurun_out/ws_$m.json || exit 1
  python3 - $m <<'PY'
import json, sys
d = json.load(open("gpurun_out/ws_%s.json" % sys.argv[1]))
print(sys.argv[1], round(d["value"], 1), [round(1e3 * k["avg_ms_per_eval"], 1) for k in d["roofline"]["kernels"]])
PY
done
