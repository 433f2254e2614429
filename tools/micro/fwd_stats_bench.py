#!/usr/bin/env python
"""1x1 forward convs WITH the BatchNorm-statistics epilogue (what the Bottlenecks run), TFLOP/s per shape.
A/B through DCFP_IGEMM_P128_STATS=0/1 (256-row tiles, one workgroup per CU / 128-row tiles, two per CU)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from dcfp_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
for (N, Cin, H, W, Cout) in [(4, 256, 128, 256, 1024), (4, 128, 128, 256, 512), (4, 64, 256, 512, 256), (4, 512, 128, 256, 2048),
                             (4, 1024, 128, 256, 256)]:
    x = torch.randn(N, Cin, H, W, device=dev)
    w = torch.randn(Cout, Cin, 1, 1, device=dev) * Cin ** -0.5
    fn = lambda: ops.conv2d_fwd(x, w, None, 1, 0, 1, want_stats=True)
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        fn()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 20
    print(f"{Cin:5d} -> {Cout:5d} @ {N}x{H}x{W}: {ms:7.3f} ms  {2.0 * N * H * W * Cin * Cout / ms / 1e9:6.1f} TF", flush=True)
