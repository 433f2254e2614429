#!/bin/bash
# usage: tools/prof_k256.sh <outdir>     (on the GPU box, from the repo root)
# Counters of the persistent 1x1 kernel on the K = 256 layers against a K = 1024 layer: SQ busy / wait
# split, MFMA busy, clock, and the L2's memory-side request / stall counters - each group in its own pass.
out=$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
args="tools/conv_bench.py --shapes l3c3_1x1,l3c1_1x1,l4c3_1x1 --passes fwd,dgrad --iters 4"
rocprofv3 --kernel-trace --output-format csv -d $out/p1 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA -- python $args > $out/p1.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $out/p2 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE -- python $args > $out/p2.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $out/p3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum -- python $args > $out/p3.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $out/p4 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_TAG_STALL_sum -- python $args > $out/p4.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $out/p5 --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum -- python $args > $out/p5.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $out/p6 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum -- python $args > $out/p6.log 2>&1
tail -2 $out/p1.log $out/p3.log $out/p4.log $out/p5.log $out/p6.log
python tools/pmc_summary.py $out igemm2_dma1p > $out/summary.txt 2>&1
cat $out/summary.txt
