#!/bin/bash
# time bench.py with each gptools_amd/libvar_*.so swapped in for libgptools_hip.so (tuning experiments)
cd "$GRAFT_REPO_ROOT"
cp gptools_amd/libgptools_hip.so /tmp/lib_real.so
for f in gptools_amd/libvar_*.so; do
  cp $f gptools_amd/libgptools_hip.so
  timeout -k 10 120 python bench.py --steps 5 --warmup 2 --batch 4 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline()); print('$f', '%.0f evals/s' % r['value'], ' '.join('%s=%.1fus' % (k['name'].split()[0], 1e3*k['avg_ms_per_eval']) for k in r['roofline']['kernels']))" || echo "$f failed"
done
cp /tmp/lib_real.so gptools_amd/libgptools_hip.so
