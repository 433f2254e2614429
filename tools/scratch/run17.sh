cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests/test_model_gpu.py tests/test_ops_gpu.py tests/test_pruner_host.py tests/test_ddp_gpu.py tests/test_ddp2_gpu.py tests/test_misc_random_gpu.py -x -q -m gpu > gpurun_out/tests_rest.txt 2>&1
echo rc=$? >> gpurun_out/tests_rest.txt
