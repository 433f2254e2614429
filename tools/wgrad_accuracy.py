#!/usr/bin/env python
"""Error of the wgrad kernels against an fp64 CPU reference on a long reduction
(K = 4*128*256 = 131072 pixels): exact-fp32 MFMA path vs the experimental 3-way bf16 split."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def run(mode):
    os.environ["DCFP_CONV_MATH"] = mode
    from dcfp_amd import ops
    g = torch.Generator().manual_seed(0)
    N, C, H, W, K, d = 4, 256, 128, 256, 3, 2
    x = torch.randn(N, C, H, W, generator=g)
    x = torch.relu(x) + 0.1 * x              # post-ReLU-like, non-zero mean
    dy = torch.randn(N, 256, H, W, generator=g) * 1e-3
    dw, _ = ops.conv2d_wgrad(dy.cuda(), x.cuda(), (256, C, K, K), 1, d, d)
    # fp64 reference on a slice of output channels / input channels (CPU time)
    xs, dys = x[:, :32].double(), dy[:, :16].double()
    w = torch.zeros(16, 32, K, K, dtype=torch.float64, requires_grad=True)
    y = torch.nn.functional.conv2d(xs, w, None, 1, d, d)
    y.backward(dys)
    ref = w.grad
    got = dw[:16, :32].double().cpu()
    rel = ((got - ref).norm() / ref.norm()).item()
    mx = ((got - ref).abs().max() / ref.abs().max()).item()
    print(f"{mode:8s} rel-L2 {rel:.3e}  max-norm {mx:.3e}")


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1])
    else:
        for m in ("f32", "bf16x3"):
            subprocess.run([sys.executable, __file__, m], check=True)
