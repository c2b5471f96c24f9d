#!/bin/bash
# run scripts/trace_col.py with each gptools_amd/libvar_trace*.so swapped in (debug builds with -DGPT_TRACE=1)
cd "$GRAFT_REPO_ROOT"
cp gptools_amd/libgptools_hip.so /tmp/lib_real.so
for f in gptools_amd/libvar_trace*.so; do
  cp $f gptools_amd/libgptools_hip.so
  echo "== $f"
  timeout -k 10 120 python scripts/trace_col.py 2>&1 | grep -E "workgroups|tile|wg |packed|forward" || echo failed
done
cp /tmp/lib_real.so gptools_amd/libgptools_hip.so
