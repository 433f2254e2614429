cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests/ -x -q -m gpu > gpurun_out/tests_all.txt 2>&1
echo rc=$? >> gpurun_out/tests_all.txt
