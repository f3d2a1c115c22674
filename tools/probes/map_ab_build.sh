#!/bin/bash
# here: tools/probes/ab/<name>.so = the library with map_kernels.hip replaced by the given source file
# usage: map_ab_build.sh <name> <map_kernels variant .hip> [extra hipcc flags]
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
name=$1; src=$2; shift 2
B=$R/alfalfa_amd/csrc/build
cp "$src" $R/alfalfa_amd/csrc/_ab_variant.hip
trap 'rm -f $R/alfalfa_amd/csrc/_ab_variant.hip /tmp/_ab_variant.o' EXIT
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c $R/alfalfa_amd/csrc/_ab_variant.hip -o /tmp/_ab_variant.o
OBJS=$(ls $B/*.o | grep -v map_kernels.hip.o)
mkdir -p $R/tools/probes/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/probes/ab/$name.so $OBJS /tmp/_ab_variant.o -lz
echo built $name
