"""Randomised (seeded) parity of the HBM-bound kernels: BatchNorm(+ReLU)(+residual) forward /
backward on odd shapes (vector and scalar paths, tiny and large channel populations), max-pool,
and the fused bilinear-upsample + cross-entropy at non-integer ratios — against fp64 torch."""
import random

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rel(a, b):
    a = a.double().cpu(); b = b.double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def _bn_cases(n, seed):
    rng = random.Random(seed)
    return [(rng.choice([1, 2, 5]), rng.choice([1, 3, 19, 64, 95, 257]), rng.randint(1, 40), rng.randint(1, 67),
             rng.random() < 0.6, rng.random() < 0.4) for _ in range(n)]


@pytest.mark.parametrize("case", _bn_cases(24, 77))
def test_random_bn(cuda, case):
    from dcfp_amd import ops
    N, C, H, W, relu, res = case
    if N * H * W < 2:
        pytest.skip("BatchNorm needs more than one value per channel in training mode")
    g = torch.Generator().manual_seed(hash(case) & 0xffff)
    x = torch.randn(N, C, H, W, generator=g) * 1.7 + 0.3
    gamma = torch.rand(C, generator=g) + 0.5
    beta = torch.randn(C, generator=g) * 0.2
    r = torch.randn(N, C, H, W, generator=g) if res else None
    dy = torch.randn(N, C, H, W, generator=g)
    xx = x.double().requires_grad_(True); gg = gamma.double().requires_grad_(True)
    bb = beta.double().requires_grad_(True)
    rr = r.double().requires_grad_(True) if res else None
    rm, rv = torch.zeros(C, dtype=torch.float64), torch.ones(C, dtype=torch.float64)
    y = F.batch_norm(xx, rm, rv, gg, bb, True, 0.1, 1e-5)
    if res:
        y = y + rr
    if relu:
        y = F.relu(y)
    y.backward(dy.double())
    xg = x.to(cuda).requires_grad_(True); gmg = gamma.to(cuda).requires_grad_(True)
    bg = beta.to(cuda).requires_grad_(True)
    rg = r.to(cuda).requires_grad_(True) if res else None
    rmg, rvg = torch.zeros(C, device=cuda), torch.ones(C, device=cuda)
    yg = ops.batch_norm_act(xg, gmg, bg, rmg, rvg, rg, relu, True, 0.1, 1e-5, False)
    yg.backward(dy.to(cuda))
    torch.cuda.synchronize()
    assert (yg.double().cpu() - y).abs().max().item() < 3e-5
    assert (rmg.double().cpu() - rm).abs().max().item() < 1e-6
    assert (rvg.double().cpu() - rv).abs().max().item() < 1e-5 * max(1.0, rv.abs().max().item())
    # ReLU masks can flip where |pre-activation| ~ 1e-7: judge gradients by norm
    assert rel(xg.grad, xx.grad) < 2e-4
    assert rel(gmg.grad, gg.grad) < 2e-4 and rel(bg.grad, bb.grad) < 2e-4
    if res:
        assert rel(rg.grad, rr.grad) < 2e-4


def _ce_cases(n, seed):
    rng = random.Random(seed)
    out = []
    for _ in range(n):
        h, w = rng.randint(2, 20), rng.randint(2, 24)
        out.append((rng.choice([1, 2, 3]), rng.choice([2, 7, 19, 33]), h, w,
                    rng.randint(h, 8 * h + 3), rng.randint(w, 8 * w + 5), rng.random() < 0.5))
    return out


@pytest.mark.parametrize("case", _ce_cases(16, 5))
def test_random_upsample_ce(cuda, case):
    from dcfp_amd import ops
    N, C, h, w, H, W, align = case
    g = torch.Generator().manual_seed(hash(case) & 0xffff)
    z = torch.randn(N, C, h, w, generator=g) * 2.5
    lab = torch.randint(0, C, (N, H, W), generator=g)
    lab[torch.rand(N, H, W, generator=g) < 0.15] = 255
    zr = z.double().requires_grad_(True)
    loss = F.cross_entropy(F.interpolate(zr, size=(H, W), mode="bilinear", align_corners=align), lab,
                           ignore_index=255)
    loss.backward()
    zg = z.to(cuda).requires_grad_(True)
    lg = ops.upsample_cross_entropy(zg, lab.to(cuda), (H, W), align, 255)
    lg.backward()
    assert abs(lg.item() - loss.item()) < 3e-6 * max(1.0, abs(loss.item()))
    assert rel(zg.grad, zr.grad) < 3e-5


@pytest.mark.parametrize("shape", [(1, 3, 7, 9), (2, 64, 33, 47), (3, 17, 2, 2), (1, 5, 64, 1),
                                   (2, 5, 9, 12), (1, 3, 6, 1028), (2, 4, 16, 4),    # W % 4 == 0: the 4-column backward
                                   (2, 3, 8, 8), (1, 2, 7, 2056), (1, 4, 5, 16), (3, 2, 1, 24)])   # W % 8 == 0: the 4-output forward
def test_random_maxpool(cuda, shape):
    from dcfp_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(shape, generator=g)
    xr = x.double().requires_grad_(True)
    y = F.max_pool2d(xr, 3, 2, 1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy.double())
    xg = x.to(cuda).requires_grad_(True)
    yg = ops.maxpool3x3s2(xg)
    yg.backward(dy.to(cuda))
    assert torch.equal(yg.cpu(), y.float())
    assert (xg.grad.double().cpu() - xr.grad).abs().max().item() < 1e-6
