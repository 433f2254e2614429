#!/bin/bash
# A/B of the persistent 1x1 kernel variants (one process per variant; DCFP_LIB selects the build)
out=gpurun_out/$1; mkdir -p $out
S="--shapes l3c3_1x1,l3c1_1x1,l4c3_1x1 --passes fwd,dgrad --iters 20"
run() { name=$1; shift; env "$@" python tools/conv_bench.py $S 2>&1 | grep -v amdgpu.ids > $out/ab_$name.log; }
L=$PWD/dcfp_amd
run base DCFP_LIB=$L/libdcfp_hip.so
for n in 1 2 3 4 6; do run stag$n DCFP_LIB=$L/libdcfp_hip_dbg.so DCFP_DBG_P=$((n*256)); done
run base2 DCFP_LIB=$L/libdcfp_hip.so
for f in $out/ab_*.log; do echo "== $f"; cat $f; done
