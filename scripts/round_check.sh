#!/bin/bash
# GPU box: full gpu test suite, smoke, bench (with cpu baseline) and a rocprofv3 kernel-trace summary.
# usage: scripts/round_check.sh <tag>
tag=${1:-check}
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag; mkdir -p $out
timeout -k 10 400 python -u -m pytest tests -m gpu -q --timeout 300 --timeout-method thread > $out/tests.log 2>&1; echo "tests exit=$?"; tail -3 $out/tests.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke exit=$?"; tail -4 $out/smoke.log
timeout -k 10 400 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench exit=$?"; cut -c1-400 $out/bench.json
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 bench.py --no-cpu-baseline > $out/prof_bench.json 2> $out/prof.err; echo "rocprof exit=$?"
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); cp "$f" $out/kernel_stats.csv; head -8 $out/kernel_stats.csv | cut -c1-200
