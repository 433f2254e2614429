#!/bin/bash
# Round evidence on one MI355X (run through gpurun from the repo root): bench lines, rocprofv3 kernel stats of the same
# command, PMC traffic / SQ counters of the fused Winograd kernel, A/B of the fused kernel against the three passes.
# usage: tools/collect_evidence.sh <tag>      -> gpurun_out/<tag>_*
tag=${1:-r04}
cd $GRAFT_REPO_ROOT
out=gpurun_out
timeout -k 10 400 python bench.py --steps 8 --warmup 3 > $out/${tag}_final_bench_n1.json 2> $out/${tag}_final_bench_n1.err || exit 1
( cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof -- python bench.py --steps 8 --warmup 3 --no-cpu-baseline > $out/${tag}_final_bench_n1_under_rocprof.json 2> $out/${tag}_prof.err ) || exit 1
find $out/${tag}_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/${tag}_final_bench_n1_kernel_stats.csv
rm -rf $out/${tag}_prof
timeout -k 10 300 python bench.py --steps 6 --warmup 3 --force-ddp --no-cpu-baseline > $out/${tag}_bench_ddp_ws1.json 2> $out/${tag}_bench_ddp_ws1.err || echo "ddp rehearsal failed"
timeout -k 10 300 python bench.py --steps 6 --warmup 3 --force-ddp --syncbn-p2p --no-cpu-baseline > $out/${tag}_bench_ddp_ws1_p2p.json 2> $out/${tag}_bench_ddp_ws1_p2p.err || echo "ddp p2p rehearsal failed"
timeout -k 10 300 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --channel-cfg tools/data/channel_cfg_r101_p60_synthetic.pth > $out/${tag}_bench_pruned_cfg5.json 2> $out/${tag}_bench_pruned.err || echo "pruned bench failed"
bash tools/prof_traffic.sh $out/${tag}_traffic tools/conv_bench.py --shapes l3c2_3x3d2 --passes fwd,dgrad --pitched --iters 4 > /dev/null 2>&1
python tools/pmc_summary.py $out/${tag}_traffic wino > $out/${tag}_winograd_traffic_pmc.txt 2>&1
rm -rf $out/${tag}_traffic
bash tools/prof_pmc.sh $out/${tag}_sq tools/conv_bench.py --shapes l3c2_3x3d2,l4c2_3x3d4,aspp_3x3d12 --passes fwd,wgrad --pitched --iters 4 > /dev/null 2>&1
python tools/pmc_summary.py $out/${tag}_sq wino_fused wino_wgrad_fused > $out/${tag}_wino_fused_sq_pmc.txt 2>&1
rm -rf $out/${tag}_sq
# HBM-side traffic of the fused Winograd weight gradient and of the stem kernels (FETCH_SIZE / WRITE_SIZE passes)
bash tools/prof_traffic.sh $out/${tag}_traffic2 tools/conv_bench.py --shapes l3c2_3x3d2,l4c2_3x3d4 --passes wgrad --pitched --iters 4 > /dev/null 2>&1
python tools/pmc_summary.py $out/${tag}_traffic2 wino_wgrad_fused wino_dw_reduce wino_wg_table > $out/${tag}_wino_wgrad_traffic_pmc.txt 2>&1
rm -rf $out/${tag}_traffic2
timeout -k 10 200 python tools/infer_bench.py --batch 4 --iters 10 > $out/${tag}_infer.txt 2>&1
timeout -k 10 200 python tools/infer_bench.py --batch 1 --iters 20 >> $out/${tag}_infer.txt 2>&1
echo done
