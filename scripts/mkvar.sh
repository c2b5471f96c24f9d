#!/bin/bash
# build gptools_amd/libvar_<name>.so: the library with the 4096-point kernels recompiled with extra flags
# usage: scripts/mkvar.sh <name> "<extra hipcc flags>"      (run `make -C gptools_amd/csrc` first)
set -e
name=$1; extra=$2
cd "$(dirname "$0")/../gptools_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wno-unused-function $extra -DGPT_N=4096 -c gpt_pow2_n.hip -o /tmp/var_$name.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wno-unused-function $extra -c gpt_pow2.hip -o /tmp/var_${name}_disp.o
objs=$(ls *.o | grep -v -e '^p2_4096.o$' -e '^gpt_pow2.o$')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libvar_$name.so $objs /tmp/var_${name}_disp.o /tmp/var_$name.o
echo built libvar_$name.so
