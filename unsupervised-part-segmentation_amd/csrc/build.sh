#!/bin/bash
# Build libupsparts_hip.so for gfx950 (cross-compiles without a GPU).  Usage: build.sh [outdir]
set -e
cd "$(dirname "$0")"
OUT=${1:-.}
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -Wno-inline-asm"
mkdir -p build
pids=()
for f in conv_igemm conv3x3_patch conv3x3_first conv3x3_s2 conv3x3_rows conv_wgrad conv_wgrad3x3 conv_wgrad3x3_f8 conv_aux pointwise partpath priors latent_adam critic; do
  if [ ! -f build/$f.o ] || [ $f.hip -nt build/$f.o ] || [ common.h -nt build/$f.o ] || [ tile.h -nt build/$f.o ] || [ build.sh -nt build/$f.o ] || [ ../../include/upsparts_hip.h -nt build/$f.o ]; then
    $HIPCC $FLAGS -c $f.hip -o build/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC build/*.o -o $OUT/libupsparts_hip.so
echo "built $OUT/libupsparts_hip.so"
