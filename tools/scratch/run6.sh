cd $GRAFT_REPO_ROOT
for F in 0 1; do echo "FUSED=$F"; DCFP_WINO_FUSED=$F timeout -k 10 250 python tools/micro/wino_fused_variants.py l3c2_3x3d2,l4c2_3x3d4,l4c2_3x3d16,aspp_3x3d12,aspp_3x3d24,ds_3x3 2>&1 | grep -v "MIOpen\|amdgpu.ids"; done > gpurun_out/fused_variants.txt 2>&1
