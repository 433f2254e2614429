"""Fan-in dgrad with / without the BatchNorm-sums side output (layer3 conv1: 1024 <- 256 @4x128x256), against the separate
bn_bwd_reduce it replaces.  usage: python tools/micro/fanin_red_bench.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from dcfp_amd import ops

dev = torch.device("cuda", 0)
torch.manual_seed(0)
N, Cin, Cout, H, W = 4, 1024, 256, 128, 256
xs = (N, Cin, H, W)
dy = torch.randn(N, Cout, H, W, device=dev)
w = torch.randn(Cout, Cin, 1, 1, device=dev) * 0.05
fan_src = torch.randn(xs, device=dev)
c3 = torch.randn(xs, device=dev) * 1.5 + 0.3
res = torch.randn(xs, device=dev)
g, b = torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev) * 0.1
mean, var = ops.bn_stats(c3)
_, mask = ops.bn_apply_relu_mask(c3, mean, var, g, b, 1e-5, res)
c3b = torch.randn(xs, device=dev) * 0.7 - 0.2            # the PREVIOUS block's bn3 input, mask, mean
mean_b, var_b = ops.bn_stats(c3b)
_, mask_b = ops.bn_apply_relu_mask(c3b, mean_b, var_b, g, b, 1e-5, res)
slots = ops.conv2d_dgrad_fanin_red_slots(w, xs)
print("slots", slots)
dx0 = ops.conv2d_dgrad_fanin(dy, w, xs, fan_src, mask)
dx1, part = ops.conv2d_dgrad_fanin_red(dy, w, xs, fan_src, mask, c3b, mask_b, mean_b, slots)
print("dx bit-equal", bool(torch.equal(dx0, dx1)))
s1, s2, dg = ops.bn_bwd_sums_from_partials(part, var_b, 1e-5)
r1, r2, rg = ops.bn_bwd_reduce(dx0, c3b, mask_b, mean_b, var_b, g, b, 1e-5, 3)
gm = dx0.double() * ((ops_bits := None) or 1)            # fp64 reference from the mask bits
def rel(a, t): return float((a.double() - t.double()).norm() / t.double().norm())
print("sum_dy vs reduce kernel %.2e   sum_dy_xmu %.2e   dgamma %.2e" % (rel(s1, r1), rel(s2, r2), rel(dg, rg)))
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / n
print("fanin            %.3f ms" % t(lambda: ops.conv2d_dgrad_fanin(dy, w, xs, fan_src, mask)))
print("fanin + sums     %.3f ms" % t(lambda: ops.conv2d_dgrad_fanin_red(dy, w, xs, fan_src, mask, c3b, mask_b, mean_b, slots)))
print("finalize         %.3f ms" % t(lambda: ops.bn_bwd_sums_from_partials(part, var_b, 1e-5)))
print("bn_bwd_reduce    %.3f ms" % t(lambda: ops.bn_bwd_reduce(dx0, c3b, mask_b, mean_b, var_b, g, b, 1e-5, 3)))
