#!/bin/bash
# Generic three-pass path (scripts/bench_generic.py, all its shapes): the library as built, the build whose column tiles
# are numbered in grid order (libvar_noxcd.so: -DGPT_GEN_XCD_TILES=0), and 8192-element column tiles (GPT_GEN_COL_TILE).
out=gpurun_out/ab_generic; mkdir -p $out
run() { # label env...
  local label=$1; shift
  env "$@" timeout -k 10 300 python3 scripts/bench_generic.py > $out/$label.jsonl 2> $out/$label.err || { echo "$label failed"; tail -3 $out/$label.err; exit 1; }
  python3 - $label $out/$label.jsonl <<'PY'
import json, sys
for l in open(sys.argv[2]):
    r = json.loads(l)
    print(sys.argv[1], r["shape"], r["chains_per_launch"], "us/launch %.1f" % r["us_per_launch"], "GB/s %.0f" % r["GBps_algorithmic"])
PY
}
run default GPT_DUMMY=0
run grid_order_tiles GPT_LIB=$PWD/gptools_amd/libvar_noxcd.so
run col_tile_8192 GPT_GEN_COL_TILE=8192
