#!/bin/bash
# run bench with each ablation library swapped in for libgptools_hip.so (timing-only experiments)
cd "$GRAFT_REPO_ROOT"
cp gptools_amd/libgptools_hip.so /tmp/lib_real.so
for ab in 0 1 2 3 4 7; do
  if [ $ab != 0 ]; then cp gptools_amd/libgptools_ab$ab.so gptools_amd/libgptools_hip.so; fi
  timeout -k 10 120 python bench.py --steps 5 --warmup 2 --batch 4 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.readline()); print('ablate $ab:', ' '.join('%s=%.1fus' % (k['name'].split()[0], 1e3*k['avg_ms_per_eval']) for k in r['roofline']['kernels']))" || echo "ablate $ab failed"
  cp /tmp/lib_real.so gptools_amd/libgptools_hip.so
done
