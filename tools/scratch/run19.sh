cd $GRAFT_REPO_ROOT
DCFP_CONV_WINOGRAD=0 timeout -k 10 800 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "not bf16x3" > gpurun_out/tests_w0.txt 2>&1
echo rc=$? >> gpurun_out/tests_w0.txt
