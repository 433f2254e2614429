cd $GRAFT_REPO_ROOT
for WV in 4 8; do echo "DCFP_WF_WAVES=$WV"; DCFP_WF_WAVES=$WV timeout -k 10 250 python tools/micro/wino_fused_variants.py l3c2_3x3d2,l4c2_3x3d4,ds_3x3,stem2_3x3,aspp_3x3d12 2>&1 | grep -v "MIOpen\|amdgpu.ids"; done > gpurun_out/waves8_ab.txt
DCFP_WF_WAVES=8 timeout -k 10 500 python -m pytest tests/test_winograd_gpu.py -x -q -k "not still_covered" > gpurun_out/tests_w8.txt 2>&1
echo rc=$? >> gpurun_out/tests_w8.txt
