cd $GRAFT_REPO_ROOT
DCFP_WINO_FUSED=1 timeout -k 10 250 python tools/micro/wino_fused_variants.py l3c2_3x3d2,l4c2_3x3d4,ds_3x3,stem2_3x3 2>&1 | grep -v "MIOpen\|amdgpu.ids" > gpurun_out/lag2.txt
timeout -k 10 1100 python -m pytest tests/ -x -q -m gpu > gpurun_out/tests_all.txt 2>&1
echo rc=$? >> gpurun_out/tests_all.txt
