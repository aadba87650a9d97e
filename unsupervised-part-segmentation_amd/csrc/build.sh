#!/bin/bash
# Build libupsparts_hip.so for gfx950 (cross-compiles without a GPU).  Usage: build.sh [outdir]
# Files with listing gates (flags.sh: ups_file_gates) are also compiled to a device listing, which tools/check_listing.py must pass
# before the library is linked: a rebuild with another hipcc cannot silently bring back a form this tree has measured wrong.
set -e
cd "$(dirname "$0")"
OUT=${1:-.}
. ./flags.sh
B=${UPS_BUILD_DIR:-build}        # (object directory: an A/B build of the whole library keeps its own)
mkdir -p $B $OUT
pids=()
gated=()
for f in $UPS_SOURCES; do
  if [ ! -f $B/$f.o ] || [ $f.hip -nt $B/$f.o ] || [ common.h -nt $B/$f.o ] || [ tile.h -nt $B/$f.o ] || [ build.sh -nt $B/$f.o ] || [ flags.sh -nt $B/$f.o ] || [ ../../include/upsparts_hip.h -nt $B/$f.o ]; then
    ups_quiet $HIPCC $UPS_FLAGS $(ups_file_flags $f) -c $f.hip -o $B/$f.o &
    pids+=($!)
    if [ -n "$(ups_file_gates $f)" ]; then
      ups_quiet $HIPCC $UPS_FLAGS $(ups_file_flags $f) -S --cuda-device-only $f.hip -o $B/$f.s &
      pids+=($!)
      gated+=($f)
    fi
  fi
done
for p in "${pids[@]}"; do wait $p; done
for f in "${gated[@]}"; do
  python3 ../../tools/check_listing.py --rules "$(ups_file_gates $f)" $B/$f.s || { rm -f $B/$f.o; echo "listing gate failed for $f.hip: not linking"; exit 1; }
done
$HIPCC --offload-arch=gfx950 -shared -fPIC $B/*.o -o $OUT/libupsparts_hip.so
$HIPCC --version | head -1 > $OUT/libupsparts_hip.hipcc_version
echo "built $OUT/libupsparts_hip.so"
