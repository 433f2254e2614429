#!/usr/bin/env python
"""Micro-benchmark of the conv C-ABI entry points on the FLOP-dominant shapes of
DeepLabv3-R101 @ 4x3x1024x2048.  Prints TFLOP/s per pass; variants
of a kernel are A/B-ed in ONE process via DCFP_* environment switches read by the library."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from dcfp_amd import ops  # noqa: E402

SHAPES = {
    # name: (N, Cin, H, W, Cout, k, stride, pad, dil)
    "l3c2_3x3d2": (4, 256, 128, 256, 256, 3, 1, 2, 2),
    "l3c3_1x1": (4, 256, 128, 256, 1024, 1, 1, 0, 1),
    "l3c1_1x1": (4, 1024, 128, 256, 256, 1, 1, 0, 1),
    "aspp_3x3d12": (4, 2048, 128, 256, 256, 3, 1, 12, 12),
    "aspp_3x3d24": (4, 2048, 128, 256, 256, 3, 1, 24, 24),
    "aspp_3x3d36": (4, 2048, 128, 256, 256, 3, 1, 36, 36),
    "l4c2_3x3d16": (4, 512, 128, 256, 512, 3, 1, 16, 16),
    "ds_3x3": (4, 1024, 128, 256, 512, 3, 1, 1, 1),
    "l4c2_3x3d4": (4, 512, 128, 256, 512, 3, 1, 4, 4),
    "l4c3_1x1": (4, 512, 128, 256, 2048, 1, 1, 0, 1),
    "l4c1_1x1": (4, 2048, 128, 256, 512, 1, 1, 0, 1),
    "stem2_3x3": (4, 64, 512, 1024, 64, 3, 1, 1, 1),
    "stem3_3x3": (4, 64, 512, 1024, 128, 3, 1, 1, 1),
    "l1c2_3x3": (4, 64, 256, 512, 64, 3, 1, 1, 1),
    "l2c2_3x3": (4, 128, 128, 256, 128, 3, 1, 1, 1),
    "l2c2_3x3s2": (4, 128, 256, 512, 128, 3, 2, 1, 1),      # layer2.0 conv2 (stride 2)
    "l2ds_1x1s2": (4, 256, 256, 512, 512, 1, 2, 0, 1),      # layer2.0 downsample (stride 2)
    "l1c3_1x1": (4, 64, 256, 512, 256, 1, 1, 0, 1),
    "l1c1_1x1": (4, 256, 256, 512, 64, 1, 1, 0, 1),
    "l2c3_1x1": (4, 128, 128, 256, 512, 1, 1, 0, 1),
    "l2c1_1x1": (4, 512, 128, 256, 128, 1, 1, 0, 1),
    # layer3 block after a 60 %-FLOPs prune (tools/pipeline_cfg5.sh): widths off every tile grid
    "p_l3c1_1x1": (4, 1024, 128, 256, 236, 1, 1, 0, 1),
    "p_l3c2_3x3d2": (4, 236, 128, 256, 232, 3, 1, 2, 2),
    "p_l3c3_1x1": (4, 232, 128, 256, 1024, 1, 1, 0, 1),
    "p_l3c2b": (4, 204, 128, 256, 188, 3, 1, 2, 2),
    "p_l3c2c": (4, 169, 128, 256, 147, 3, 1, 2, 2),
    "p_last0": (4, 512, 128, 256, 154, 3, 1, 1, 1),
    "p_aspp83": (4, 2048, 128, 256, 83, 3, 1, 36, 36),
    "p_aspp57": (4, 2048, 128, 256, 57, 3, 1, 24, 24),
    "p_aspp40": (4, 2048, 128, 256, 40, 3, 1, 12, 12),
    "p_stem": (4, 62, 512, 1024, 118, 3, 1, 1, 1),
    "p_l3c1b": (4, 1024, 128, 256, 210, 1, 1, 0, 1),
}


def bench(fn, iters):
    for _ in range(max(3, iters)):      # the clocks take tens of ms to ramp after an idle gap
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="l3c2_3x3d2,l3c3_1x1,l3c1_1x1,aspp_3x3d12,ds_3x3")
    ap.add_argument("--passes", default="fwd,dgrad,wgrad")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--check", action="store_true", help="compare all three passes with torch's GPU conv on the full shape")
    ap.add_argument("--pitched", action="store_true",
                    help="x / dy row-pitched with a zero tail where the library supports it (ops.conv_pitch)")
    ap.add_argument("--zeros", action="store_true",
                    help="all-zero operands: same instruction stream at much lower switching power")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    for name in args.shapes.split(","):
        N, Cin, H, W, Cout, k, s, p, d = SHAPES[name]
        x = torch.randn(N, Cin, H, W, device=dev)
        w = torch.randn(Cout, Cin, k, k, device=dev) * (Cin * k * k) ** -0.5
        y = ops.conv2d_fwd(x, w, None, s, p, d)
        dy = torch.randn_like(y)
        if args.zeros:
            x.zero_(); w.zero_(); dy.zero_()
        if args.pitched:
            pitch = ops.conv_pitch(tuple(x.shape), tuple(w.shape), s, p, d)
            if pitch:
                xp = ops.new_pitched(tuple(x.shape), pitch, dev); xp.copy_(x); x = xp
                dp = ops.new_pitched(tuple(dy.shape), pitch, dev); dp.copy_(dy); dy = dp
            name = name + ("*" if pitch else "")
        flops = 2.0 * N * Cout * y.shape[2] * y.shape[3] * Cin * k * k
        res = {}
        if "fwd" in args.passes:
            res["fwd"] = bench(lambda: ops.conv2d_fwd(x, w, None, s, p, d), args.iters)
        if "dgrad" in args.passes:
            res["dgrad"] = bench(lambda: ops.conv2d_dgrad(dy, w, tuple(x.shape), s, p, d), args.iters)
        if "wgrad" in args.passes:
            res["wgrad"] = bench(lambda: ops.conv2d_wgrad(dy, x, tuple(w.shape), s, p, d), args.iters)
        line = f"{name:14s} " + "  ".join(f"{k_}: {v:7.3f} ms {flops / v / 1e9:6.1f} TF" for k_, v in res.items())
        print(line, flush=True)
        if args.check:   # full shape against torch's own (MIOpen, fp32) convolution on the GPU
            def rel(a, b):
                return ((a.double() - b.double()).norm() / b.double().norm()).item()
            xr = x.clone().requires_grad_(True)
            wr = w.clone().requires_grad_(True)
            yr = torch.nn.functional.conv2d(xr, wr, None, s, p, d)
            gx, gw = torch.autograd.grad(yr, (xr, wr), dy)
            print(f"   vs torch: fwd {rel(ops.conv2d_fwd(x, w, None, s, p, d), yr):.2e}"
                  f"  dgrad {rel(ops.conv2d_dgrad(dy, w, tuple(x.shape), s, p, d), gx):.2e}"
                  f"  wgrad {rel(ops.conv2d_wgrad(dy, x, tuple(w.shape), s, p, d)[0], gw):.2e}", flush=True)

if __name__ == "__main__":
    main()
