"""Randomised conv parity (seeded): shapes drawn across every tile configuration of the forward /
data-gradient / weight-gradient kernels — channel counts off every grid, odd image sizes, strides,
dilations, padding smaller and larger than the dilation reach, accumulate-dgrad — against fp64 CPU
convolutions.  Complements the fixed cases of test_ops_gpu.py (reference layer shapes) and
test_conv_large_gpu.py (256 x 256-tile kernels)."""
import math
import os
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


# lop-sided weight-gradient tiles (conv_wgrad.hip cfg 3 / 4: 256 x 128, 256 x 64), 1x1 and 3x3,
# with ragged M / N and the kernel each is expected to route to
LOPSIDED = [
    ((2, 128, 33, 40, 512, 1, 1, 0, 1), "wgrad2_kernel<1,4,2,2,2>"),   # layer2 conv3-like: M 512, Nn 128
    ((2, 100, 20, 28, 300, 1, 1, 0, 1), "wgrad2_kernel<1,4,2,2,2>"),   # ragged M and Nn
    ((2, 64, 40, 48, 256, 1, 1, 0, 1), "wgrad2_kernel<1,4,1,2,2>"),    # layer1 conv3: M 256, Nn 64
    ((1, 47, 17, 23, 150, 1, 1, 0, 1), "wgrad2_kernel<1,4,1,2,2>"),
    ((2, 256, 24, 36, 64, 1, 1, 0, 1), "wgrad2_kernel<1,2,2,1,1>"),    # layer1 conv1: M 64, Nn 256 (one-wave tile)
    ((2, 64, 30, 44, 64, 3, 1, 1, 1), "wgrad2_kernel<9,2,2,1,1>"),     # stem / layer1 3x3: M 64, Nn 576
    ((2, 12, 21, 33, 200, 3, 1, 2, 2), "wgrad2_kernel<9,4,2,2,2>"),    # Nn 108
    ((2, 5, 25, 31, 256, 3, 1, 1, 1), "wgrad2_kernel<9,4,1,2,2>"),     # Nn 45
]


@pytest.mark.parametrize("case,kernel", LOPSIDED)
def test_lopsided_wgrad_tiles(cuda, case, kernel):
    from dcfp_amd import ops, _lib
    N, Cin, H, W, Cout, k, s, p, d = case
    desc = ops._desc((N, Cin, H, W), (Cout, Cin, k, k), s, p, d)
    assert ops.conv_kernel_name(desc, _lib.CONV_WGRAD) == kernel
    g = torch.Generator().manual_seed(17)
    x = torch.randn(N, Cin, H, W, generator=g)
    w64 = (torch.randn(Cout, Cin, k, k, generator=g).double()).requires_grad_(True)
    y64 = F.conv2d(x.double(), w64, None, s, p, d)
    dy = torch.randn(y64.shape, generator=g)
    y64.backward(dy.double())
    dw, db = ops.conv2d_wgrad(dy.to(cuda), x.to(cuda), tuple(w64.shape), s, p, d, need_bias=True)
    assert rel(dw, w64.grad) < 2e-5, rel(dw, w64.grad)
    assert rel(db, dy.double().sum((0, 2, 3))) < 1e-5


# stride-2 data gradients: tiles hold one output phase (h % 2, w % 2) and run only the taps that reach it; a 1x1
# conv reaches one phase, whose tiles write the zeros of the other three (vector form when W is a multiple of 8)
STRIDED = [(2, 64, 32, 48, 96, 1, 2, 0, 1), (1, 32, 33, 50, 40, 1, 2, 0, 1), (2, 20, 16, 24, 300, 1, 2, 0, 1),
           (2, 16, 16, 16, 24, 3, 2, 1, 1), (1, 16, 17, 19, 24, 3, 2, 1, 1), (1, 8, 20, 24, 16, 3, 2, 2, 2),
           (1, 8, 21, 25, 16, 3, 2, 0, 1), (2, 130, 30, 34, 140, 3, 2, 1, 1), (1, 12, 18, 18, 8, 3, 3, 1, 1)]


@pytest.mark.parametrize("case", STRIDED)
def test_strided_dgrad_phases(cuda, case):
    from dcfp_amd import ops
    N, Cin, H, W, Cout, k, s, p, d = case
    g = torch.Generator().manual_seed(23)
    x64 = torch.randn(N, Cin, H, W, generator=g).double().requires_grad_(True)
    w = torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)
    y64 = F.conv2d(x64, w.double(), None, s, p, d)
    dy = torch.randn(y64.shape, generator=g)
    y64.backward(dy.double())
    dx = torch.full((N, Cin, H, W), float("nan"), device=cuda)          # every element must be written
    ops.conv2d_dgrad(dy.to(cuda), w.to(cuda), (N, Cin, H, W), s, p, d, out=dx, accumulate=False)
    assert torch.isfinite(dx).all()
    assert rel(dx, x64.grad) < 1e-5, rel(dx, x64.grad)
    seed = torch.randn(N, Cin, H, W, generator=g)
    dxa = seed.to(cuda)
    ops.conv2d_dgrad(dy.to(cuda), w.to(cuda), (N, Cin, H, W), s, p, d, out=dxa, accumulate=True)
    assert rel(dxa, x64.grad + seed.double()) < 1e-5


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


# the Cin = 3 stride-2 stem conv (networks/backbone/resnet.py:88-90) on its own kernels (dcfp_amd/csrc/conv_stem.hip: no LDS,
# operands straight into the MFMA lane layout): forward (+ bias), weight gradient (wave slabs reduced in a fixed order) against
# fp64; image borders (row / column -1, odd heights: the last patch row outside), widths that leave a partly filled wave
# (a wave takes 128 output pixels), fewer than 3 input channels, a dy that is a batch-strided view, and the BatchNorm
# statistics epilogue (output widths that are multiples of 128) against the statistics kernel on the same output
STEM = [(2, 3, 64, 128), (1, 3, 51, 64), (3, 2, 33, 192), (2, 3, 128, 512), (4, 1, 16, 64), (1, 3, 24, 1280)]


@pytest.mark.parametrize("case", STEM)
def test_stem_conv_kernels(cuda, case):
    import torch
    import torch.nn.functional as F
    from dcfp_amd import ops, _lib
    N, Cin, H, W = case
    g = torch.Generator().manual_seed(23)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(64, Cin, 3, 3, generator=g) / (3.0 * Cin ** 0.5)
    b = torch.randn(64, generator=g)
    desc = ops._desc(x.shape, w.shape, 2, 1, 1)
    on = os.environ.get("DCFP_CONV_STEM", "1") != "0"
    assert (ops.conv_kernel_name(desc, _lib.CONV_FWD) == "stem_fwd_kernel") == on
    assert (ops.conv_kernel_name(desc, _lib.CONV_WGRAD) == "stem_wgrad_kernel") == on
    xd, wd = x.to(cuda), w.to(cuda)
    y = ops.conv2d_fwd(xd, wd, None, 2, 1, 1)
    yb = ops.conv2d_fwd(xd, wd, b.to(cuda), 2, 1, 1)
    ref = F.conv2d(x.double(), w.double(), None, 2, 1, 1)
    assert tuple(y.shape) == tuple(ref.shape)
    assert ((y.cpu().double() - ref).abs().max() / ref.abs().max()).item() < 2e-6
    assert ((yb.cpu().double() - (ref + b.double().view(1, -1, 1, 1))).abs().max() / ref.abs().max()).item() < 2e-6
    r = ops.conv2d_fwd(xd, wd, None, 2, 1, 1, want_stats=True)
    ys, st = r if isinstance(r, tuple) else (r, None)
    assert torch.equal(ys, y)
    assert (st is not None) == (on and ref.shape[3] % 128 == 0), (st is None, ref.shape)
    if st is not None:
        m_ref, v_ref = ops.bn_stats(y)
        assert float((st[0] - m_ref).abs().max() / m_ref.abs().max()) < 2e-5
        assert float(((st[1] - v_ref).abs() / v_ref).max()) < 2e-5
    dy = torch.randn(ref.shape, generator=g)
    dw = ops.conv2d_wgrad(dy.to(cuda), xd, tuple(w.shape), 2, 1, 1)[0]
    dw2 = ops.conv2d_wgrad(dy.to(cuda), xd, tuple(w.shape), 2, 1, 1)[0]
    refw = torch.nn.grad.conv2d_weight(x.double(), w.shape, dy.double(), 2, 1, 1)
    assert torch.equal(dw, dw2)                                    # fixed summation order
    assert ((dw.cpu().double() - refw).norm() / refw.norm()).item() < 2e-5
    wide = torch.randn(N, 96, *ref.shape[2:], generator=g).to(cuda)      # dy as a channel slice: its own image stride
    wide[:, 16:80] = dy.to(cuda)
    assert torch.equal(ops.conv2d_wgrad(wide[:, 16:80], xd, tuple(w.shape), 2, 1, 1)[0], dw)
