cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_conv_large_gpu.py tests/test_winograd_gpu.py -x -q -k "not still_covered" > gpurun_out/tests_p.txt 2>&1
echo rc=$? >> gpurun_out/tests_p.txt
