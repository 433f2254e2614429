#!/usr/bin/env python
"""Timing of the Winograd forward variants the training step runs (plain / keep V / statistics / both) and of the dgrad
variants (plain / accumulate) on the R101 shapes: A/B of DCFP_WINO_FUSED in two processes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dcfp_amd import ops
from tools.conv_bench import SHAPES, bench

dev = torch.device("cuda:0")
for name in sys.argv[1].split(","):
    N, Cin, H, W, Cout, k, s, p, d = SHAPES[name]
    x = torch.randn(N, Cin, H, W, device=dev)
    w = torch.randn(Cout, Cin, k, k, device=dev) * (Cin * k * k) ** -0.5
    dy = torch.randn(N, Cout, H, W, device=dev)
    pitch = ops.conv_pitch(tuple(x.shape), tuple(w.shape), s, p, d)
    if pitch:
        xp = ops.new_pitched(tuple(x.shape), pitch, dev); xp.copy_(x); x = xp
        dp = ops.new_pitched(tuple(dy.shape), pitch, dev); dp.copy_(dy); dy = dp
    y = torch.empty(N, Cout, H, W, device=dev)
    dx = torch.zeros(N, Cin, H, W, device=dev)
    r = {}
    r["fwd"] = bench(lambda: ops.conv2d_fwd(x, w, None, s, p, d, out=y), 10)
    r["fwd+keep"] = bench(lambda: ops.conv2d_fwd(x, w, None, s, p, d, out=y, keep={}), 10)
    r["fwd+stats"] = bench(lambda: ops.conv2d_fwd(x, w, None, s, p, d, out=y, want_stats=True), 10)
    r["fwd+keep+stats"] = bench(lambda: ops.conv2d_fwd(x, w, None, s, p, d, out=y, want_stats=True, keep={}), 10)
    r["dgrad"] = bench(lambda: ops.conv2d_dgrad(dy, w, tuple(x.shape), s, p, d, out=dx), 10)
    r["dgrad+acc"] = bench(lambda: ops.conv2d_dgrad(dy, w, tuple(x.shape), s, p, d, out=dx, accumulate=True), 10)
    kp = {}
    ops.conv2d_fwd(x, w, None, s, p, d, out=y, keep=kp)
    r["wgrad"] = bench(lambda: ops.conv2d_wgrad(dy, x, tuple(w.shape), s, p, d), 10)
    r["wgrad_kept"] = bench(lambda: ops.conv2d_wgrad(dy, x, tuple(w.shape), s, p, d, xform=kp.get("xform")), 10)
    print(f"{name:12s}" + "  ".join(f"{k_}: {v:6.3f}" for k_, v in r.items()), flush=True)
