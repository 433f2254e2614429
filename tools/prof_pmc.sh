#!/bin/bash
# usage: tools/prof_pmc.sh <outdir> <python args...>   (run on the GPU box, from the repo root)
# Two PMC passes over the same command (counters in their own runs, kernel-trace only).
out=$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $out/p1 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA -- python "$@" > $out/p1.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $out/p2 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE -- python "$@" > $out/p2.log 2>&1
ls -R $out | head -30
