cd $GRAFT_REPO_ROOT
for P in 1 0; do echo "DCFP_WF_PERSIST=$P"; DCFP_WF_PERSIST=$P timeout -k 10 250 python tools/micro/wino_fused_variants.py l3c2_3x3d2,l4c2_3x3d4,ds_3x3,stem2_3x3,l2c2_3x3 2>&1 | grep -v "MIOpen\|amdgpu.ids"; done > gpurun_out/persist_ab.txt
timeout -k 10 600 python -m pytest tests/test_winograd_gpu.py tests/test_conv_large_gpu.py -x -q -k "not still_covered" > gpurun_out/tests_p.txt 2>&1
echo rc=$? >> gpurun_out/tests_p.txt
