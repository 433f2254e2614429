set -x
cd $GRAFT_REPO_ROOT
S=l3c2_3x3d2,l4c2_3x3d4,l4c2_3x3d16,aspp_3x3d12,ds_3x3
for F in 0 1; do
  DCFP_WINO_FUSED=$F timeout -k 10 200 python tools/conv_bench.py --shapes $S --passes fwd,dgrad --pitched --check --iters 10 > gpurun_out/fused_bench_$F.txt 2>&1 || exit 1
done
DCFP_WINO_FUSED=1 timeout -k 10 200 python tools/conv_bench.py --shapes $S --passes fwd,dgrad --check --iters 10 > gpurun_out/fused_bench_dense.txt 2>&1 || exit 1
timeout -k 10 500 python -m pytest tests/test_winograd_gpu.py -x -q -k vs_fp64 > gpurun_out/fused_test.txt 2>&1
echo rc=$?
