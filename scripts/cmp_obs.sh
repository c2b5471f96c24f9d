#!/bin/bash
# GPU box: time the non-centred model density (scripts/bench_obs.py, normal family only) with each
# gptools_amd/libvar_*.so against the built library
cd "$GRAFT_REPO_ROOT"
for f in gptools_amd/libgptools_hip.so gptools_amd/libvar_*.so; do
  [ -f "$f" ] || continue
  tag=$(basename $f .so)
  GPT_LIB=$PWD/$f OBS_NEED_F=0 OBS_FAMILIES=normal timeout -k 10 200 python scripts/bench_obs.py 2>/dev/null | cut -c40-70,230-420 | sed "s/^/$tag /" || exit 1
done
