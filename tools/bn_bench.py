#!/usr/bin/env python
"""GB/s of the four BatchNorm kernels on the two tensor sizes that dominate DeepLabv3-R101 at
4x3x1024x2048 (layer3 block: 1024 and 256 channels at 128x256).  Algorithmic bytes as DESIGN.md §3."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from dcfp_amd import ops  # noqa: E402


def bench(fn, iters=10):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    dev = torch.device("cuda:0")
    for shape in [(4, 1024, 128, 256), (4, 256, 128, 256), (4, 2048, 128, 256)]:
        x = torch.randn(shape, device=dev)
        dy = torch.randn(shape, device=dev)
        res = torch.randn(shape, device=dev)
        C = shape[1]
        g = torch.rand(C, device=dev) + 0.5
        b = torch.randn(C, device=dev) * 0.1
        mean, var = ops.bn_stats(x)
        n = x.numel()
        y = ops.bn_apply(x, mean, var, g, b, 1e-5, res, True)
        s1, s2, _ = ops.bn_bwd_reduce(dy, x, None, mean, var, g, b, 1e-5, 2)
        rows = [
            ("stats", 4 * n, lambda: ops.bn_stats(x)),
            ("apply", 8 * n, lambda: ops.bn_apply(x, mean, var, g, b, 1e-5, None, True)),
            ("apply+res", 12 * n, lambda: ops.bn_apply(x, mean, var, g, b, 1e-5, res, True)),
            ("bwd_reduce(x-mask)", 8 * n, lambda: ops.bn_bwd_reduce(dy, x, None, mean, var, g, b, 1e-5, 2)),
            ("bwd_reduce(y-mask)", 12 * n, lambda: ops.bn_bwd_reduce(dy, x, y, mean, var, g, b, 1e-5, 1)),
            ("bwd_apply(x-mask)", 12 * n, lambda: ops.bn_bwd_apply(dy, x, None, mean, var, g, b, 1e-5, s1, s2, float(n // C), 2, False)),
            ("bwd_apply(y-mask,+res)", 20 * n, lambda: ops.bn_bwd_apply(dy, x, y, mean, var, g, b, 1e-5, s1, s2, float(n // C), 1, True)),
        ]
        def fwd_pair():
            m_, v_ = ops.bn_stats(x)
            ops.bn_apply(x, m_, v_, g, b, 1e-5, None, True)

        def bwd_pair():
            a_, b_, _ = ops.bn_bwd_reduce(dy, x, None, mean, var, g, b, 1e-5, 2)
            ops.bn_bwd_apply(dy, x, None, mean, var, g, b, 1e-5, a_, b_, float(n // C), 2, False)
        # the sequences the step runs: the second kernel re-reads what the first just streamed
        rows += [("stats -> apply", 12 * n, fwd_pair), ("bwd_reduce -> bwd_apply", 20 * n, bwd_pair)]
        print(shape, "DCFP_BN_ORDER=" + os.environ.get("DCFP_BN_ORDER", "0"))
        for name, nbytes, fn in rows:
            ms = bench(fn)
            print(f"  {name:24s} {ms:7.3f} ms  {nbytes / ms / 1e6:7.0f} GB/s", flush=True)


if __name__ == "__main__":
    main()
