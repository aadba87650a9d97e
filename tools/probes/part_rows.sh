python3 -m pytest tests/test_gpu_kernels.py -q -m gpu -k "unpool or mask_parts" 2>&1 | tail -5
for sh in 64,128,10 32,256,16 16,256,20 64,128,25; do python3 tools/hbm_roofline.py --shape $sh --only mask_parts_fwd,unpool_bwd 2>&1 | grep -v "^$" | tail -4; done
