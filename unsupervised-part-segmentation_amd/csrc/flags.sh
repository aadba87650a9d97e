# Compiler flags of libupsparts_hip.so, sourced by build.sh and by the A/B builders (tools/ab_build.sh, tools/asm_patch_build.sh) so that
# an A/B library differs from the shipped one by exactly what its command line says.
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
UPS_FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -Wno-inline-asm"
UPS_SOURCES="conv_igemm conv3x3_patch conv3x3_first conv3x3_s2 conv3x3_rows conv_wgrad conv_wgrad3x3 conv_wgrad3x3_f8 conv_aux pointwise partpath priors latent_adam critic"
# per-file flags.  EVERY translation unit is compiled WITHOUT packed fp32 VALU instructions: rule 1 of conv3x3_rows.hip's header
# (docs/design/rows_hazard.md: the one wrong result this tree has measured from correct code was a v_pk_add_f32 beside a sibling
# wave's MFMA section), extended to the whole library as insurance because it costs nothing -- in the MFMA files hipcc's packed forms
# spilled (conv3x3_patch 14 -> 0 spilled registers, conv_wgrad3x3 110 -> 0) and the step is 1.0 .. 2.1 % FASTER without them
# (profiles/round6_nopk_all.txt); in the HBM-bound files, whose waves share SIMDs with other streams' MFMA kernels, nothing changes
# (same file: 2 019.5 / 2 019.6 against 2 026 / 2 013 img/s, every row of the HBM table within 3 %).  The feature switch is a cc1 option, so
# the HOST pass of the same command line sees it too and says "not a recognized feature": filtered.
# (A/B: UPS_NOPK_FILES="conv3x3_rows" UPS_BUILD_DIR=build_x build.sh ab/x builds a whole library with another set)
UPS_NOPK_FILES=${UPS_NOPK_FILES:-$UPS_SOURCES}
ups_file_flags() {
  case " $UPS_NOPK_FILES " in
    *" $1 "*) [ "$1" = conv3x3_rows ] && [ -n "$UPS_ROWS_ALLOW_PK" ] || echo "-Xclang -target-feature -Xclang -packed-fp32-ops" ;;
  esac
}
ups_quiet() { "$@" 2> >(grep -v "is not a recognized feature for this target" >&2); }
# listing gates (tools/check_listing.py <rules> <listing>): which rules a file's device listing must pass before the library is linked
ups_file_gates() {
  case "$1" in
    conv3x3_rows) echo "no-packed-fp32 asm-loads" ;;
    *) case " $UPS_NOPK_FILES " in *" $1 "*) echo "no-packed-fp32" ;; esac ;;
  esac
}
