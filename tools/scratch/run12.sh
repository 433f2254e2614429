cd $GRAFT_REPO_ROOT
S=p_l3c2_3x3d2,p_l3c2b,p_l3c2c,p_last0,p_aspp83,p_aspp57,p_aspp40,p_stem
for F in 0 1; do echo "DCFP_WINO_FUSED=$F"; DCFP_WINO_FUSED=$F timeout -k 10 250 python tools/conv_bench.py --shapes $S --passes fwd,dgrad --pitched --iters 10 2>&1 | grep -v "MIOpen\|amdgpu.ids"; done > gpurun_out/pruned_ab.txt 2>&1
timeout -k 10 1100 python -m pytest tests/ -x -q -m gpu > gpurun_out/tests_all.txt 2>&1
echo rc=$? >> gpurun_out/tests_all.txt
