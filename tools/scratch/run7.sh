cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_winograd_gpu.py -x -q -k vs_fp64 > gpurun_out/fused_test.txt 2>&1
echo test rc=$? >> gpurun_out/fused_test.txt
DCFP_WINO_FUSED=1 timeout -k 10 250 python tools/micro/wino_fused_variants.py l3c2_3x3d2,l4c2_3x3d4,l4c2_3x3d16,aspp_3x3d12,aspp_3x3d24,ds_3x3 2>&1 | grep -v "MIOpen\|amdgpu.ids" > gpurun_out/fused_variants1.txt
