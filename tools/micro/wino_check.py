"""Winograd F(2x2,3x3) path against fp64 (small shapes) and against the direct kernels (error ratio, time).
Run twice: DCFP_CONV_WINOGRAD=0 / 1 (the library reads the switch once)."""
import os, sys, json, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dcfp_amd import ops, _lib

dev = torch.device("cuda:0")
out = {}
# accuracy: (N, Cin, H, W, Cout, d)
for (N, Cin, H, W, Cout, d) in [(2, 256, 64, 128, 256, 2), (2, 256, 50, 68, 512, 1), (1, 512, 64, 128, 256, 12),
                                (2, 256, 33, 60, 256, 4), (2, 320, 64, 128, 256, 36)]:
    g = torch.Generator().manual_seed(3)
    x = torch.randn(N, Cin, H, W, generator=g); w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
    dy = torch.randn(N, Cout, H, W, generator=g)
    xd, wd, dyd = x.to(dev), w.to(dev), dy.to(dev)
    desc = ops._desc(x.shape, w.shape, 1, d, d)
    name = ops.conv_kernel_name(desc, _lib.CONV_FWD)
    y = ops.conv2d_fwd(xd, wd, None, 1, d, d)
    dx = ops.conv2d_dgrad(dyd, wd, tuple(x.shape), 1, d, d)
    seed = torch.randn(x.shape, generator=g)
    acc = seed.to(dev).clone()
    ops.conv2d_dgrad(dyd, wd, tuple(x.shape), 1, d, d, out=acc, accumulate=True)
    dw, _ = ops.conv2d_wgrad(dyd, xd, tuple(w.shape), 1, d, d)
    wname = ops.conv_kernel_name(desc, _lib.CONV_WGRAD)
    torch.cuda.synchronize()
    refdw = torch.nn.grad.conv2d_weight(x.double(), w.shape, dy.double(), 1, d, d)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), None, 1, d, d)
    refdx = torch.nn.grad.conv2d_input(x.shape, w.double(), dy.double(), 1, d, d)
    e = lambda a, b: float((a.cpu().double() - b).abs().max() / b.abs().max())
    r = lambda a, b: float((a.cpu().double() - b).norm() / b.norm())
    out[f"{N}x{Cin}x{H}x{W}->{Cout} d{d}"] = {"kernel": name, "fwd_max": e(y, ref), "fwd_rel": r(y, ref),
                                               "dgrad_max": e(dx, refdx), "dgrad_rel": r(dx, refdx),
                                               "acc_max": e(acc, refdx + seed.double()),
                                               "wgrad_kernel": wname, "wgrad_max": e(dw, refdw), "wgrad_rel": r(dw, refdw),
                                               "frac": ops.conv_executed_fraction(desc, _lib.CONV_FWD)}
print("WINO " + json.dumps(out, indent=1))
