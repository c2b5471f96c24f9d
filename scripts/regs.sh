#!/bin/bash
# Register / spill report of the kernels of one translation unit: scripts/regs.sh <file.hip> [hipcc flags...]
# prints: kernel  VGPRs  AGPRs  scratch-bytes/lane  occupancy  VGPR-spills
f=$1; shift
cd "$(dirname "$0")/../gptools_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc "$@" -Rpass-analysis=kernel-resource-usage -c "$f" -o /tmp/regs_$$.o 2>&1 |
  awk '/Function Name:/ {name=$(NF-1)} / VGPRs:/ {v=$(NF-1)} /AGPRs:/ {a=$(NF-1)} /ScratchSize/ {s=$(NF-1)} /Occupancy/ {o=$(NF-1)} /VGPRs Spill/ {print name, "vgpr=" v, "agpr=" a, "scratch=" s, "occ=" o, "spill=" $(NF-1)}' |
  sed -e 's/\[-Rpass-analysis=kernel-resource-usage\]//' | c++filt | sed -e 's/(LatArgs.*) / /' -e 's/void pow2:://'
rm -f /tmp/regs_$$.o
