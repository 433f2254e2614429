"""How much of a memory-bound BatchNorm backward hides behind a weight-gradient GEMM on a second stream?
(layer3 sizes of DeepLabv3-R101 at 4x3x1024x2048).  Serial time of both vs both streams at once."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dcfp_amd import ops

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
# BN backward on the 1024-channel stream (bn3) and on a 256-channel tensor
x3 = torch.randn(4, 1024, 128, 256, device=dev); dy3 = torch.randn_like(x3)
gam = torch.rand(1024, device=dev) + 0.5; bet = torch.zeros(1024, device=dev)
mean, var = ops.bn_stats(x3)
# weight gradients: 1x1 256->1024 (conv3), 1x1 1024->256 (conv1), Winograd 3x3 256->256 d2 (conv2)
xa = torch.randn(4, 256, 128, 256, device=dev); dya = torch.randn(4, 1024, 128, 256, device=dev)
xb = torch.randn(4, 256, 128, 256, device=dev); dyb = torch.randn(4, 256, 128, 256, device=dev)


def bn_bwd():
    s1, s2, _ = ops.bn_bwd_reduce(dy3, x3, None, mean, var, gam, bet, 1e-5, 2)
    ops.bn_bwd_apply(dy3, x3, None, mean, var, gam, bet, 1e-5, s1, s2, float(x3.numel() // 1024), 2, False)


def wgrads():
    ops.conv2d_wgrad(dya, xa, (1024, 256, 1, 1), 1, 0, 1)
    ops.conv2d_wgrad(dyb, xb, (256, 256, 3, 3), 1, 2, 2)


def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


side = torch.cuda.Stream()


def both():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        wgrads()
    bn_bwd(); bn_bwd()
    main.wait_stream(side)


with torch.cuda.stream(side):
    wgrads()            # side-stream workspaces
torch.cuda.synchronize()
t_bn = timed(lambda: (bn_bwd(), bn_bwd()))
t_wg = timed(wgrads)
t_both = timed(both)
print(f"2 x bn3 backward {t_bn:.3f} ms   wgrads (1x1 + Winograd 3x3) {t_wg:.3f} ms   serial {t_bn + t_wg:.3f} ms   two streams {t_both:.3f} ms")
