cd ${GRAFT_REPO_ROOT:-.}
python3 -m pytest tests/test_gpu_kernels.py -q -m gpu -k "priors or unpool or mask_parts or softmax or moments" 2>&1 | tail -3
python3 -m pytest tests/test_gpu_model.py -q -m gpu -x 2>&1 | tail -3
O=gpurun_out/profiles_round6; mkdir -p $O; TAG=round6
(for sh in 64,128,10 32,256,16 16,256,20 64,128,25; do python3 tools/hbm_roofline.py --shape $sh 2>&1 | grep -v amdgpu.ids; echo; done
 echo "---- the protocol of rounds 1-5 (one operand set re-used by every launch), P = 10:"; python3 tools/hbm_roofline.py --same-buffers 2>&1 | grep -v amdgpu.ids
 echo "---- UPS_PRIOR_DIRECT=0 (the staged prior kernels of round 4 at P = 16 / 20 / 25):"
 for sh in 32,256,16 16,256,20 64,128,25; do UPS_PRIOR_DIRECT=0 python3 tools/hbm_roofline.py --shape $sh --only prior 2>&1 | grep -v amdgpu.ids; done) > $O/${TAG}_hbm_kernels.txt
python3 tools/hbm_roofline.py --json $O/${TAG}_hbm_kernels.json > /dev/null 2>&1
for cf in deepfashion256p16 cub256p20; do python3 bench.py --no-cpu-baseline --config $cf 2>/dev/null | grep '"metric"' > $O/${TAG}_bench_$cf.json; done
python3 bench.py --no-cpu-baseline --config cub256p20 --precision bf16 2>/dev/null | grep '"metric"' > $O/${TAG}_bench_cub256p20_bf16.json
