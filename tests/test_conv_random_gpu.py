"""Randomised conv parity (seeded): shapes drawn across every tile configuration of the forward /
data-gradient / weight-gradient kernels — channel counts off every grid, odd image sizes, strides,
dilations, padding smaller and larger than the dilation reach, accumulate-dgrad — against fp64 CPU
convolutions.  Complements the fixed cases of test_ops_gpu.py (reference layer shapes) and
test_conv_large_gpu.py (256 x 256-tile kernels)."""
import math
import random

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _cases(n, seed):
    rng = random.Random(seed)
    out = []
    while len(out) < n:
        k = rng.choice([1, 3])
        s = rng.choice([1, 1, 1, 2])
        d = rng.choice([1, 1, 2, 3, 6]) if k == 3 else 1
        p = rng.choice([0, d, d * (k // 2), d + 1]) if k == 3 else 0
        N = rng.choice([1, 2, 3])
        Cin = rng.choice([3, 16, 17, 47, 64, 95, 130, 200, 257])
        Cout = rng.choice([8, 19, 33, 64, 95, 128, 150, 256, 300])
        H, W = rng.randint(5, 48), rng.randint(5, 72)
        ho = (H + 2 * p - d * (k - 1) - 1) // s + 1
        wo = (W + 2 * p - d * (k - 1) - 1) // s + 1
        if ho < 1 or wo < 1:
            continue
        out.append((N, Cin, H, W, Cout, k, s, p, d))
    return out


def rel(a, b):
    a = a.double().cpu(); b = b.double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


@pytest.mark.parametrize("case", _cases(36, 20240607))
def test_random_conv(cuda, case):
    from dcfp_amd import ops
    N, Cin, H, W, Cout, k, s, p, d = case
    g = torch.Generator().manual_seed(hash(case) & 0xffff)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)
    x64 = x.double().requires_grad_(True); w64 = w.double().requires_grad_(True)
    y64 = F.conv2d(x64, w64, None, s, p, d)
    dy = torch.randn(y64.shape, generator=g)
    y64.backward(dy.double())
    xg, wg, dyg = x.to(cuda), w.to(cuda), dy.to(cuda)
    y = ops.conv2d_fwd(xg, wg, None, s, p, d)
    dx = ops.conv2d_dgrad(dyg, wg, tuple(x.shape), s, p, d)
    seed = torch.randn(x.shape, generator=g)
    dxa = seed.to(cuda)
    ops.conv2d_dgrad(dyg, wg, tuple(x.shape), s, p, d, out=dxa, accumulate=True)
    dw = ops.conv2d_wgrad(dyg, xg, tuple(w.shape), s, p, d)[0]
    torch.cuda.synchronize()
    K = Cin * k * k
    tol = 3e-6 * max(1.0, math.sqrt(K) / 8)
    assert tuple(y.shape) == tuple(y64.shape)
    assert rel(y, y64) < tol, ("fwd", case, rel(y, y64))
    assert rel(dx, x64.grad) < max(tol, 1e-5), ("dgrad", case, rel(dx, x64.grad))
    assert rel(dxa, x64.grad + seed.double()) < max(tol, 1e-5), ("dgrad+acc", case)
    assert rel(dw, w64.grad) < 2e-5, ("wgrad", case, rel(dw, w64.grad))
