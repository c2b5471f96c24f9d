#!/bin/bash
# GPU box: rocprofv3 kernel statistics of scripts/bench_generic.py, one shape per run (the generic three-pass path's kernels)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/gen_prof; mkdir -p $out
for s in "$@"; do
  timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$s -o p -- python3 scripts/bench_generic.py $s > $out/$s.log 2>&1 || { echo "$s failed"; tail -5 $out/$s.log; exit 1; }
  tail -1 $out/$s.log | cut -c1-200
  python3 - $out/$s <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:6]:
    print("   %-70s calls %5s avg %10.1f us  %5s %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
done
