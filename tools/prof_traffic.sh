#!/bin/bash
# HBM traffic of the conv kernels (separate --pmc passes: FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2).
out=$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $out/f --pmc FETCH_SIZE -- python "$@" > $out/f.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $out/w --pmc WRITE_SIZE -- python "$@" > $out/w.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $out/h --pmc TCC_HIT_sum TCC_MISS_sum -- python "$@" > $out/h.log 2>&1
