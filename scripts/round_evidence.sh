#!/bin/bash
# GPU box: smoke, the default bench line (with cpu baseline), the rocprofv3 kernel-trace summary of the same command,
# and the FETCH_SIZE / WRITE_SIZE passes at the bench's unit size. Every step under its own limit; the first failure ends it.
# usage: scripts/round_evidence.sh <tag>
tag=${1:-evidence}
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/$tag; mkdir -p "$out"
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.log" 2>&1 || { echo "smoke failed"; tail -5 "$out/smoke.log"; exit 1; }
tail -2 "$out/smoke.log"
timeout -k 10 420 python bench.py > "$out/bench.json" 2> "$out/bench.err" || { echo "bench failed"; tail -5 "$out/bench.err"; exit 1; }
cut -c1-600 "$out/bench.json"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof" -- python3 bench.py --no-cpu-baseline --no-extra > "$out/prof_bench.json" 2> "$out/prof.err" || { echo "rocprof failed"; tail -5 "$out/prof.err"; exit 1; }
f=$(find "$out/prof" -name "*kernel_stats.csv" | head -1); cp "$f" "$out/kernel_stats.csv"; head -8 "$out/kernel_stats.csv" | cut -c1-200
PMC_GROUPS="FETCH_SIZE;WRITE_SIZE;SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
  bash scripts/pmc_passes.sh "$out/pmc" --batch 16 --no-extra > "$out/pmc_summary.txt" 2>&1 || { echo "pmc failed"; tail -5 "$out/pmc_summary.txt"; exit 1; }
tail -30 "$out/pmc_summary.txt"
