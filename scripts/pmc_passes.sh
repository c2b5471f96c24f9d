#!/bin/bash
# Collect rocprofv3 PMC counters for the bench kernels, one counter group per run (gfx950 slot limits).
# usage: scripts/pmc_passes.sh <outdir> [bench args...]
set -u
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
i=0
if [ -n "${PMC_GROUPS:-}" ]; then IFS=';' read -ra groups <<< "$PMC_GROUPS"; else groups=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "GRBM_GUI_ACTIVE"); fi
for grp in "${groups[@]}"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $grp --output-format csv -d "$out/p$i" -- python3 bench.py --no-cpu-baseline --no-profile --steps 2 --warmup 1 --batch 2 "$@" > "$out/p$i.log" 2>&1 || { echo "pass $i ($grp) failed"; tail -3 "$out/p$i.log"; exit 1; }
done
python3 scripts/pmc_summary.py "$out"
