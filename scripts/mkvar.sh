#!/bin/bash
# build gptools_amd/libvar_<name>.so: the library with the 4096-point kernels recompiled with extra flags
# usage: scripts/mkvar.sh <name> "<extra hipcc flags>"      (run `make -C gptools_amd/csrc` first)
set -e
name=$1; extra=$2
cd "$(dirname "$0")/../gptools_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wno-unused-function $extra -DGPT_N=4096 -c gpt_pow2_n.hip -o /tmp/var_$name.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wno-unused-function $extra -c gpt_pow2.hip -o /tmp/var_${name}_disp.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libvar_$name.so gpt_api.o gpt_small.o gpt_elem.o /tmp/var_${name}_disp.o gpt_generic.o p2_256.o p2_512.o p2_1024.o p2_2048.o p2_8192.o p2_16384.o /tmp/var_$name.o
echo built libvar_$name.so
