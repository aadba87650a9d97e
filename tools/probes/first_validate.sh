cd ${GRAFT_REPO_ROOT:-.}
python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_repeat.py -q -m gpu -k "first_layer or masked or sign or conv or residual or bilinear or pointwise" 2>&1 | tail -3
python3 tools/bench_first.py 2>&1 | grep -v amdgpu
python3 tools/bench_conv.py --post --f16 --bits --only dv_rb128,dv_rb64 --iters 20 2>&1 | grep dv_rb
