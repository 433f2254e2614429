#!/bin/bash
# Round-4 final evidence (second half of the round): bench lines, rocprofv3 kernel stats of the same command, the
# data-parallel rehearsals, config 5, inference, the per-entry conv listing.  usage: tools/collect_evidence_b.sh <tag>
tag=${1:-r04}
cd $GRAFT_REPO_ROOT
out=gpurun_out
timeout -k 10 500 python bench.py --steps 8 --warmup 3 > $out/${tag}_final_bench_n1.json 2> $out/${tag}_final_bench_n1.err || exit 1
( cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof -- python bench.py --steps 8 --warmup 3 --no-cpu-baseline > $out/${tag}_final_bench_n1_under_rocprof.json 2> $out/${tag}_prof.err ) || exit 1
find $out/${tag}_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/${tag}_final_bench_n1_kernel_stats.csv
rm -rf $out/${tag}_prof
timeout -k 10 300 python bench.py --steps 6 --warmup 3 --force-ddp --no-cpu-baseline > $out/${tag}_bench_ddp_ws1.json 2> $out/${tag}_bench_ddp_ws1.err || echo "ddp rehearsal failed"
timeout -k 10 300 python bench.py --steps 6 --warmup 3 --force-ddp --syncbn-p2p --no-cpu-baseline > $out/${tag}_bench_ddp_ws1_p2p.json 2> $out/${tag}_bench_ddp_ws1_p2p.err || echo "ddp p2p rehearsal failed"
timeout -k 10 300 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --channel-cfg tools/data/channel_cfg_r101_p60_synthetic.pth > $out/${tag}_bench_pruned_cfg5.json 2> $out/${tag}_bench_pruned.err || echo "pruned bench failed"
timeout -k 10 200 python tools/infer_bench.py --batch 4 --iters 10 > $out/${tag}_infer.txt 2>&1
timeout -k 10 200 python tools/infer_bench.py --batch 1 --iters 20 >> $out/${tag}_infer.txt 2>&1
timeout -k 10 200 python tools/micro/slow_layers.py --all > $out/${tag}_conv_entries.txt 2>&1
# HBM traffic of the fused BatchNorm backward (FETCH_SIZE / WRITE_SIZE passes): 12 B/element, not 20
bash tools/prof_traffic.sh $out/${tag}_bn_traffic tools/micro/bn_fused_bench.py > /dev/null 2>&1
python tools/pmc_summary.py $out/${tag}_bn_traffic bn_bwd > $out/${tag}_bn_fused_traffic_pmc.txt 2>&1
rm -rf $out/${tag}_bn_traffic
echo done
