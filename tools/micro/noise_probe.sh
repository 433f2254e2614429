mkdir -p gpurun_out/r4b
T='tests/test_model_gpu.py::test_forward_backward_vs_reference_golden'
for cfg in "default" "DCFP_CONV_WINOGRAD=0" "DCFP_BN_FUSED=0" "DCFP_FUSED_BN_STATS=0" "DCFP_CONV_WINOGRAD=0 DCFP_FUSED_BN_STATS=0 DCFP_BN_FUSED=0"; do
  echo "=== $cfg"
  if [ "$cfg" = "default" ]; then timeout -k 10 200 python -m pytest "$T" -q -m gpu -s 2>&1 | grep "gradient norms\|gradient projections\|passed\|failed"
  else env $cfg timeout -k 10 200 python -m pytest "$T" -q -m gpu -s 2>&1 | grep "gradient norms\|gradient projections\|passed\|failed"; fi
done
