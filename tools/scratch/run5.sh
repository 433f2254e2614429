cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_winograd_gpu.py -x -q -k vs_fp64 > gpurun_out/fused_test.txt 2>&1
echo test rc=$? >> gpurun_out/fused_test.txt
timeout -k 10 300 python bench.py --steps 6 --warmup 3 > gpurun_out/bench_fused.json 2> gpurun_out/bench_fused.err
echo bench rc=$?
