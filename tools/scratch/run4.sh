cd $GRAFT_REPO_ROOT
bash tools/prof_pmc.sh gpurun_out/pmc_fused tools/conv_bench.py --shapes l3c2_3x3d2,ds_3x3,l4c2_3x3d4 --passes fwd --pitched --iters 4 > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/pmc_fused wino_fused > gpurun_out/pmc_fused_summary.txt 2>&1
rm -rf gpurun_out/pmc_fused/p1 gpurun_out/pmc_fused/p2
