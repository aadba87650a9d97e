#!/bin/bash
# same-box A/B of two builds through UPS_LIB: tools/probes/lib_base.so (built from the previous tree) against the in-tree library
CASES=${1:-dv_rb128,dv_rb64,dv_rb32,ea_rb0,ea_rb1,ea_rb2,vgg1_2,vgg3_2}
for r in 1 2; do
for l in tools/probes/lib_base.so unsupervised-part-segmentation_amd/csrc/libupsparts_hip.so; do echo "== $l"; UPS_LIB=$l python3 tools/bench_conv.py --only $CASES --f16 --post --iters 10 2>&1 | grep -v amdgpu | tail -9; done; done
