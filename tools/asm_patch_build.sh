#!/bin/bash
# A/B build of the library with ONE source file's DEVICE LISTING patched between compiler and assembler -- the instrument of the
# round-6 hazard hunt (docs/design/rows_hazard.md): a hypothesis about one instruction pair is tested by fencing exactly that pair
# and nothing else, which no source-level switch can do (every source edit reshuffles hipcc's schedule).
#   tools/asm_patch_build.sh <name> <file-stem> "<-D flags>" <patch.py> [patch args ...]
#     patch.py reads the listing on stdin and writes the patched listing to stdout (tools/probes/rows_asm_patch.py)
#   (the file's shipped per-file flags of csrc/flags.sh apply; UPS_ROWS_ALLOW_PK=1 in the environment lets conv3x3_rows keep its packed
#   fp32 instructions: with "-DUPS_ROWS_FWD_SIGN -DUPS_ROWS_NO_FENCE" that is the reproducer build the hunt patched)
#   -> ab/<name>/libupsparts_hip.so (git-ignored; travels to the GPU box); use with UPS_LIB=ab/<name>/libupsparts_hip.so
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT/unsupervised-part-segmentation_amd/csrc"
N=$1; F=$2; FLAGS=$3; PATCH=$4; shift 4
LLVM=/opt/rocm/lib/llvm/bin
OUT="$ROOT/ab/$N"; mkdir -p "$OUT"
. ./flags.sh
CF="$UPS_FLAGS $(ups_file_flags $F) $FLAGS"
/opt/rocm/bin/hipcc $CF -S --cuda-device-only $F.hip -o "$OUT/$F.dev.s" 2>/dev/null
python3 "$PATCH" "$@" < "$OUT/$F.dev.s" > "$OUT/$F.patched.s"
$LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c "$OUT/$F.patched.s" -o "$OUT/$F.dev.o"
$LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared "$OUT/$F.dev.o" -o "$OUT/$F.dev.out"
$LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 \
    -input=/dev/null -input="$OUT/$F.dev.out" -output="$OUT/$F.hipfb"
ups_quiet /opt/rocm/bin/hipcc $CF --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang "$OUT/$F.hipfb" -c $F.hip -o "$OUT/$F.o"
OBJS=$(ls build/*.o | grep -v "/$F.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS "$OUT/$F.o" -o "$OUT/libupsparts_hip.so"
rm -f "$OUT/$F.dev.o" "$OUT/$F.dev.out" "$OUT/$F.hipfb" "$OUT/$F.o"
echo "built ab/$N/libupsparts_hip.so"
