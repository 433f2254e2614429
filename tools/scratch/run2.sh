cd $GRAFT_REPO_ROOT
S=l3c2_3x3d2,l4c2_3x3d4,aspp_3x3d12,ds_3x3
DCFP_WINO_FUSED=1 timeout -k 10 200 python tools/conv_bench.py --shapes $S --passes fwd,dgrad --pitched --check --iters 10 2>&1 | grep -v MIOpen > gpurun_out/fused_bench_1.txt || exit 1
DCFP_WINO_FUSED=1 timeout -k 10 200 python tools/conv_bench.py --shapes ds_3x3 --passes fwd,dgrad --check --iters 10 2>&1 | grep -v MIOpen > gpurun_out/fused_bench_dense.txt || exit 1
