cd $GRAFT_REPO_ROOT
for D in 0 1 2 3 4 7 8 15; do
echo "== dbg $D"
DCFP_WF_DBG=$D timeout -k 10 100 python tools/conv_bench.py --shapes l3c2_3x3d2,ds_3x3 --passes dgrad --pitched --iters 10 2>&1 | grep -v "MIOpen\|amdgpu.ids"
done > gpurun_out/fused_dbg.txt 2>&1
