#!/usr/bin/env python
"""Weight gradient of the 3x3 stride-1 convs of DeepLabv3-R101 through whichever kernel the
library picks (DCFP_WINO_WGRAD_FUSED=0: batched Winograd with a fresh x transform / direct; 1: the fused kernel of
conv_winograd3.hip): time per launch and relative error against fp64 on a slice of input channels.  Run once per setting
(the library reads its switches once per process)."""
import argparse
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from dcfp_amd import ops, _lib  # noqa: E402

SHAPES = {
    # name: (N, Cin, H, W, Cout, dil)
    "l3c2": (4, 256, 128, 256, 256, 2), "l4c2d4": (4, 512, 128, 256, 512, 4), "l4c2d8": (4, 512, 128, 256, 512, 8),
    "l4c2d16": (4, 512, 128, 256, 512, 16), "aspp12": (4, 2048, 128, 256, 256, 12), "aspp24": (4, 2048, 128, 256, 256, 24),
    "aspp36": (4, 2048, 128, 256, 256, 36), "ds": (4, 1024, 128, 256, 512, 1), "last0": (4, 512, 128, 256, 256, 1),
    "last3": (4, 256, 128, 256, 256, 1), "stem2": (4, 64, 512, 1024, 64, 1), "stem3": (4, 64, 512, 1024, 128, 1),
    "l1c2": (4, 64, 256, 512, 64, 1), "l2c2": (4, 128, 128, 256, 128, 1),
    "p236": (4, 236, 128, 256, 232, 2), "p204": (4, 204, 128, 256, 188, 2), "p154": (4, 512, 128, 256, 154, 1),
    "odd": (3, 100, 66, 136, 120, 2), "oddd4": (2, 96, 50, 70, 200, 4), "d12s": (2, 128, 64, 128, 128, 12),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="l3c2,l4c2d4,l4c2d16,aspp12,aspp24,last0,last3,stem2,stem3,l1c2,l2c2,p236,odd,oddd4,d12s")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--dense", action="store_true", help="do not row-pitch the dilation-1 / 2 operands")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    for name in a.shapes.split(","):
        N, Cin, H, W, Cout, d = SHAPES[name]
        g = torch.Generator().manual_seed(3)
        x = torch.randn(N, Cin, H, W, generator=g)
        x = (torch.relu(x) + 0.05 * x).to(dev)
        dy = (torch.randn(N, Cout, H, W, generator=g) * 1e-2).to(dev)
        wshape = (Cout, Cin, 3, 3)
        pitch = 0 if a.dense else ops.conv_pitch(tuple(x.shape), wshape, 1, d, d)
        xs, dys = x, dy
        if pitch:
            xs = ops.new_pitched(tuple(x.shape), pitch, dev); xs.copy_(x)
            dys = ops.new_pitched(tuple(dy.shape), pitch, dev); dys.copy_(dy)
        desc = ops._desc(x.shape, wshape, 1, d, d, pitch, pitch)
        kname = ops.conv_kernel_name(desc, _lib.CONV_WGRAD)
        dw = ops.conv2d_wgrad(dys, xs, wshape, 1, d, d)[0]
        torch.cuda.synchronize()
        nci = min(Cin, 8)
        ci = slice(Cin - nci, Cin)
        w64 = torch.zeros(Cout, nci, 3, 3, dtype=torch.float64, requires_grad=True)
        F.conv2d(x[:, ci].double().cpu(), w64, None, 1, d, d).backward(dy.double().cpu())
        ref = w64.grad.to(dev)
        rel = ((dw[:, ci].double() - ref).norm() / ref.norm()).item()
        co = slice(0, min(Cout, 4))          # first output channels against the fp32 conv of torch on the whole tensor
        for _ in range(3):
            ops.conv2d_wgrad(dys, xs, wshape, 1, d, d)
        torch.cuda.synchronize()
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(a.iters):
            ops.conv2d_wgrad(dys, xs, wshape, 1, d, d)
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / a.iters
        flops = 2.0 * N * Cout * H * W * Cin * 9
        frac = ops.conv_executed_fraction(desc, _lib.CONV_WGRAD)
        print(f"{name:8s} {'*' if pitch else ' '} {ms:7.3f} ms  {flops / ms / 1e9:6.1f} TF nominal  {flops * frac / ms / 1e9:6.1f} TF executed  "
              f"rel {rel:.2e}  finite {bool(torch.isfinite(dw).all())}  {kname}", flush=True)


if __name__ == "__main__":
    main()
