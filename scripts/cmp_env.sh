#!/bin/bash
# GPU box: headline bench under several values of one environment variable.
# usage: scripts/cmp_env.sh VAR "<bench args>" <value>...
cd "$GRAFT_REPO_ROOT"
var=$1; args=$2; shift 2
for v in "$@"; do
  env $var=$v timeout -k 10 200 python bench.py --no-cpu-baseline $args 2>/dev/null > gpurun_out/env_$v.json || exit 1
  python3 - $v <<'PY'
import json, sys
d = json.load(open("gpurun_out/env_%s.json" % sys.argv[1]))
print(sys.argv[1], round(d["value"], 1), [round(1e3 * k["avg_ms_per_eval"], 1) for k in d["roofline"]["kernels"]])
PY
done
