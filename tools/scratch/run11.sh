cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_winograd_gpu.py tests/test_ops_gpu.py tests/test_conv_random_gpu.py tests/test_conv_large_gpu.py -x -q -k "not still_covered" > gpurun_out/tests_narrow.txt 2>&1
echo rc=$? >> gpurun_out/tests_narrow.txt
timeout -k 10 300 python bench.py --steps 6 --warmup 3 > gpurun_out/bench_narrow.json 2> gpurun_out/bench_narrow.err
echo bench rc=$? >> gpurun_out/tests_narrow.txt
