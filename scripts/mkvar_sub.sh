#!/bin/bash
# build gptools_amd/libvar_<name>.so: the library with the sub-line kernels of one line length recompiled with
# extra flags.   usage: scripts/mkvar_sub.sh <name> <N> "<extra hipcc flags>"   (run make first)
set -e
name=$1; n=$2; extra=$3
cd "$(dirname "$0")/../gptools_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wno-unused-function $extra -DGPT_N=$n -c gpt_sub_n.hip -o /tmp/varsub_$name.o
objs=$(ls *.o | grep -v "^sub_$n.o" | tr '\n' ' ')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libvar_$name.so $objs /tmp/varsub_$name.o
echo built libvar_$name.so
