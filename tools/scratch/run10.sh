cd $GRAFT_REPO_ROOT
S=stem2_3x3,stem3_3x3,l1c2_3x3,l2c2_3x3
echo wino_fused; DCFP_WINO_MIN=64 DCFP_CONV_WINOGRAD=2 DCFP_WINO_FUSED=2 timeout -k 10 200 python tools/scratch/small_wino.py $S 2>&1 | grep -v "MIOpen\|amdgpu.ids"
