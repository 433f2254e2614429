#!/usr/bin/env python
"""Same-box A/B of the BatchNorm backward: two kernels (reduce 8 B/element + apply 12 B/element) against the fused
launch (12 B/element, dcfp_bn_bwd_fused_f32) on the tensor sizes of DeepLabv3-R101 at 4x3x1024x2048."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from dcfp_amd import ops  # noqa: E402


def bench(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    dev = torch.device("cuda:0")
    shapes = [(4, 256, 128, 256), (4, 512, 128, 256), (4, 1024, 128, 256), (4, 2048, 128, 256), (4, 64, 256, 512),
              (4, 256, 256, 512), (4, 128, 256, 512), (4, 64, 512, 1024), (4, 128, 512, 1024)]
    for shape in shapes:
        x = torch.randn(shape, device=dev)
        dy = torch.randn(shape, device=dev)
        C = shape[1]
        g = torch.rand(C, device=dev) + 0.5
        b = torch.randn(C, device=dev) * 0.1
        mean, var = ops.bn_stats(x)
        n = x.numel()
        cnt = float(n // C)

        def two():
            a_, b_, _ = ops.bn_bwd_reduce(dy, x, None, mean, var, g, b, 1e-5, 2)
            ops.bn_bwd_apply(dy, x, None, mean, var, g, b, 1e-5, a_, b_, cnt, 2, False)

        def red():
            ops.bn_bwd_reduce(dy, x, None, mean, var, g, b, 1e-5, 2)

        def fused():
            ops.bn_bwd_fused(dy, x, None, mean, var, g, b, 1e-5, cnt, 2, False)

        t_red, t_two, t_f = bench(red), bench(two), bench(fused)
        print(f"{shape}: reduce {t_red:6.3f} ms ({8 * n / t_red / 1e6:5.0f} GB/s)  reduce+apply {t_two:6.3f} ms "
              f"({20 * n / t_two / 1e6:5.0f} GB/s)  fused {t_f:6.3f} ms ({12 * n / t_f / 1e6:5.0f} GB/s)  x{t_two / t_f:4.2f}",
              flush=True)
        del x, dy
    torch.cuda.synchronize()
    ops.check_fused_status()


if __name__ == "__main__":
    main()
