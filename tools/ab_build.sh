#!/bin/bash
# A/B build of the library with extra -D flags on ONE source file: tools/ab_build.sh <name> <file-stem> "<flags>"
#   -> ab/<name>/libupsparts_hip.so (git-ignored; travels to the GPU box); use with UPS_LIB=ab/<name>/libupsparts_hip.so
set -e
cd "$(dirname "$0")/../unsupervised-part-segmentation_amd/csrc"
N=$1; F=$2; FLAGS=$3
mkdir -p ../../ab/$N
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -Wno-inline-asm $FLAGS -c $F.hip -o ../../ab/$N/$F.o
OBJS=$(ls build/*.o | grep -v "/$F.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS ../../ab/$N/$F.o -o ../../ab/$N/libupsparts_hip.so
echo "built ab/$N/libupsparts_hip.so"
