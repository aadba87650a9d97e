#!/bin/bash
# A/B build of the library with extra flags on ONE source file: tools/ab_build.sh <name> <file-stem> "<flags>"
#   -> ab/<name>/libupsparts_hip.so (git-ignored; travels to the GPU box); use with UPS_LIB=ab/<name>/libupsparts_hip.so
# The file's shipped per-file flags (csrc/flags.sh) apply unless the environment says otherwise (UPS_ROWS_ALLOW_PK=1: conv3x3_rows with
# packed fp32 instructions -- with "-DUPS_ROWS_FWD_SIGN -DUPS_ROWS_NO_FENCE" the reproducer of docs/design/rows_hazard.md).
set -e
cd "$(dirname "$0")/../unsupervised-part-segmentation_amd/csrc"
. ./flags.sh
N=$1; F=$2; FLAGS=$3
mkdir -p ../../ab/$N
ups_quiet $HIPCC $UPS_FLAGS $(ups_file_flags $F) $FLAGS -c $F.hip -o ../../ab/$N/$F.o
OBJS=$(ls build/*.o | grep -v "/$F.o")
$HIPCC --offload-arch=gfx950 -shared -fPIC $OBJS ../../ab/$N/$F.o -o ../../ab/$N/libupsparts_hip.so
rm -f ../../ab/$N/$F.o
echo "built ab/$N/libupsparts_hip.so"
