cd $GRAFT_REPO_ROOT
timeout -k 10 300 python bench.py --steps 6 --warmup 3 > gpurun_out/bench_fused.json 2> gpurun_out/bench_fused.err
echo bench rc=$?
