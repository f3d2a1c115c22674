#!/bin/bash
# builds tools/probes/libalfa_hot_prof.so = the library with map_kernels.hip compiled under -DALFA_MAP_PROFILE (run here, before gpurun)
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
B=$R/alfalfa_amd/csrc/build
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -Wall -Wno-unused-function -DALFA_MAP_PROFILE -c $R/alfalfa_amd/csrc/map_kernels.hip -o $R/tools/probes/map_kernels_prof.o
OBJS=$(ls $B/*.o | grep -v map_kernels.hip.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/probes/libalfa_hot_prof.so $OBJS $R/tools/probes/map_kernels_prof.o -lz
rm -f $R/tools/probes/map_kernels_prof.o
echo built
