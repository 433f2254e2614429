cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests/test_fullsize_gpu.py tests/test_model_gpu.py tests/test_ddp_gpu.py tests/test_winograd_gpu.py -x -q -k "not still_covered" > gpurun_out/tests_new.txt 2>&1
echo rc=$? >> gpurun_out/tests_new.txt
