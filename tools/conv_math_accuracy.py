#!/usr/bin/env python
"""Error of the two conv math modes against fp64, pass by pass, on value distributions that stress
a split-precision product: N(0,1), post-ReLU activations, heavy-tailed (log-normal magnitudes over
~12 decades), and globally tiny / huge scales.  The library reads DCFP_CONV_MATH once, so each mode
runs in a child process; the fp64 reference is a CPU convolution on a channel slice.
    python tools/conv_math_accuracy.py            # prints a table (profiles/r01_conv_math_accuracy.txt)"""
import json
import math
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPE = (2, 256, 128, 256, 256, 3, 2, 2)      # N, Cin, H, W, Cout, k, pad, dil  (layer3 conv2)
DISTS = ["normal", "relu", "heavy_tail", "tiny_1e-18", "huge_1e+12"]


def make(dist, shape, g):
    import torch
    t = torch.randn(shape, generator=g)
    if dist == "relu":
        t = torch.relu(t) + 0.0
    elif dist == "heavy_tail":
        t = t * torch.exp(torch.randn(shape, generator=g) * 3.0)
    elif dist.startswith("tiny"):
        t = t * 1e-18
    elif dist.startswith("huge"):
        t = t * 1e12
    return t


def child():
    import torch
    import torch.nn.functional as F
    from dcfp_amd import ops
    N, Cin, H, W, Cout, k, p, d = SHAPE
    dev = torch.device("cuda:0")
    out = {}
    for dist in DISTS:
        g = torch.Generator().manual_seed(7)
        x = make(dist, (N, Cin, H, W), g)
        w = torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)
        dy = make(dist, (N, Cout, H, W), g)
        xg, wg, dyg = x.to(dev), w.to(dev), dy.to(dev)
        y = ops.conv2d_fwd(xg, wg, None, 1, p, d)
        dx = ops.conv2d_dgrad(dyg, wg, tuple(x.shape), 1, p, d)
        dw = ops.conv2d_wgrad(dyg, xg, tuple(w.shape), 1, p, d)[0]
        torch.cuda.synchronize()
        mo, ci = slice(0, 24), slice(0, 16)
        y64 = F.conv2d(x.double(), w[mo].double(), None, 1, p, d)
        x64 = x[:, ci].double().requires_grad_(True)
        w64 = w[:, ci].double().requires_grad_(True)
        F.conv2d(x64, w64, None, 1, p, d).backward(dy.double())

        def errs(a, b):
            a = a.double().cpu()
            return [((a - b).norm() / b.norm()).item(), ((a - b).abs().max() / b.abs().max()).item()]
        out[dist] = {"fwd": errs(y[:, mo], y64), "dgrad": errs(dx[:, ci], x64.grad), "wgrad": errs(dw[:, ci], w64.grad)}
    print("ACC_RESULT " + json.dumps(out))


def main():
    if "--child" in sys.argv:
        return child()
    res = {}
    for mode in ("f32", "bf16x3"):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"],
                           env=dict(os.environ, DCFP_CONV_MATH=mode), capture_output=True, text=True, timeout=1200)
        line = [l for l in r.stdout.splitlines() if l.startswith("ACC_RESULT ")]
        if not line:
            raise SystemExit(r.stderr[-2000:])
        res[mode] = json.loads(line[-1][len("ACC_RESULT "):])
    print("layer3 conv2 shape %s; error vs fp64: rel-L2 / max-abs over max|ref|" % (SHAPE,))
    print("%-12s %-6s %-24s %-24s" % ("values", "pass", "exact fp32 MFMA", "bf16x3 split"))
    for dist in DISTS:
        for ps in ("fwd", "dgrad", "wgrad"):
            a, b = res["f32"][dist][ps], res["bf16x3"][dist][ps]
            print("%-12s %-6s %.2e / %.2e      %.2e / %.2e" % (dist, ps, a[0], a[1], b[0], b[1]))


if __name__ == "__main__":
    main()
