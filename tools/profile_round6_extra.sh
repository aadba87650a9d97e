#!/bin/bash
# usage (GPU box): UPS_TREE=<hash> bash tools/profile_round6_extra.sh <tag>
# What round 6 adds to tools/profile_round.sh: the HBM table with rotating operand sets for the four part-path shapes (and once with
# the old same-buffers protocol), graph replay against the eager step, host enqueue time, the timeline of the traced step, HBM traffic
# of the roofline layer's forward and input-gradient launches on THIS tree (UPS_TREE), the row-stream regression.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-round6}
cd $R
O=$R/gpurun_out/profiles_$TAG
mkdir -p $O
(for sh in 64,128,10 32,256,16 16,256,20 64,128,25; do python3 tools/hbm_roofline.py --shape $sh 2>&1 | grep -v amdgpu.ids; echo; done
 echo "---- the protocol of rounds 1-5 (one operand set re-used by every launch), P = 10:"; python3 tools/hbm_roofline.py --same-buffers 2>&1 | grep -v amdgpu.ids
 echo "---- UPS_PRIOR_DIRECT=0 (the staged prior kernels of round 4 at P = 16 / 20 / 25):"
 for sh in 32,256,16 16,256,20 64,128,25; do UPS_PRIOR_DIRECT=0 python3 tools/hbm_roofline.py --shape $sh --only prior 2>&1 | grep -v amdgpu.ids; done) > $O/${TAG}_hbm_kernels.txt
python3 tools/hbm_roofline.py --json $O/${TAG}_hbm_kernels.json > /dev/null 2>&1
(echo "# eager step against HIP-graph replay (no host in the loop): 40 timed steps after 20, alternating, two rounds"
 for rep in 1 2; do for g in 0 1; do echo "UPS_GRAPH=$g: $(UPS_GRAPH=$g timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 40 --warmup 20 2>/dev/null | grep metric | cut -c62-110)"; done; done
 echo "# tools/host_overhead.py"; python3 tools/host_overhead.py 2>&1 | grep -v amdgpu.ids | tail -2) > $O/${TAG}_graph_vs_eager.txt
bash tools/pmc_traffic.sh $O/${TAG}_pmc_dv_rb128.json patch_kernelIDF16_ 2215772160 -- fwd dv_rb128 bits > /dev/null 2>&1
bash tools/pmc_traffic.sh $O/${TAG}_pmc_dgrad_dv_rb128_sign_bytes.json patch_kernelIDF16b 2218131456 -- dgrad dv_rb128 bits > /dev/null 2>&1
bash tools/pmc_traffic.sh $O/${TAG}_pmc_dv_rb128_fp8.json patch_kernelIDF16b 2755002368 -- dgrad dv_rb128 f8 bits > /dev/null 2>&1
ls ab/repro/libupsparts_hip.so ab/reg_default/libupsparts_hip.so > /dev/null 2>&1 && bash tools/probes/rows_hunt.sh 800 repro reg_default default > $O/${TAG}_rows_hazard_regression.txt 2>&1
python3 tools/probes/determinism.py > $O/${TAG}_determinism.txt 2>&1
cat $O/${TAG}_graph_vs_eager.txt; head -20 $O/${TAG}_hbm_kernels.txt
