#!/bin/bash
# Builds conv3x3_patch.hip with -DUPS_PHASE_TIMING into tools/probes/lib_phase.so (run here, before gpurun) or, with `run`, executes the
# probe on the GPU box:   bash tools/probes/phase_timing.sh build ;  gpurun -- 'bash tools/probes/phase_timing.sh run "dv_rb128 fwd" ...'
cd "$(dirname "$0")/../.."
if [ "$1" = build ]; then
  C=unsupervised-part-segmentation_amd/csrc; mkdir -p /tmp/ups_dbg
  (cd $C && bash build.sh > /dev/null) && cp $C/build/*.o /tmp/ups_dbg/
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -Wno-inline-asm -DUPS_PHASE_TIMING -c $C/conv3x3_patch.hip -o /tmp/ups_dbg/conv3x3_patch.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/ups_dbg/*.o -o tools/probes/lib_phase.so && echo built tools/probes/lib_phase.so
  exit 0
fi
shift
for c in "${@:-dv_rb128 fwd}"; do
  UPS_LIB=tools/probes/lib_phase.so python3 tools/probes/phase_timing.py $c 2>&1 | grep -v amdgpu.ids
done
