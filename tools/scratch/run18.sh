cd $GRAFT_REPO_ROOT
S=l3c3_1x1,l3c1_1x1,l4c3_1x1,l4c1_1x1
for R in 0 1 0 1; do echo "DCFP_IGEMM_ROTATE=$R"; DCFP_IGEMM_ROTATE=$R timeout -k 10 200 python tools/conv_bench.py --shapes $S --passes fwd,dgrad --iters 20 2>&1 | grep -v "MIOpen\|amdgpu.ids"; done > gpurun_out/rot_ab.txt 2>&1
timeout -k 10 400 python -m pytest tests/test_conv_large_gpu.py -x -q -k "persist or bit" > gpurun_out/tests_rot.txt 2>&1
echo rc=$? >> gpurun_out/tests_rot.txt
