"""The fused BatchNorm backward (dcfp_bn_bwd_fused_f32: both stages in one launch, dy and x read once, the blocks of a
channel meeting through 8-byte tagged granules) against the two-kernel path it replaces (dcfp_bn_bwd_reduce_f32 +
dcfp_bn_bwd_apply_f32, networks/backbone/resnet.py:26-33,41-56): the same bits for every output, on every ReLU-mask
mode, ragged and multi-chunk channel populations, pitched dx, batch-strided dy, and across many calls on one hand-off
buffer (a stale granule of an earlier call must never pass for the current one's).  The two-kernel path itself is held
against fp64 torch in test_ops_gpu.py / test_misc_random_gpu.py."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _inputs(shape, dev, seed, res=True):
    g = torch.Generator().manual_seed(seed)
    N, C, H, W = shape
    x = (torch.randn(shape, generator=g) * 1.3 + 0.2).to(dev)
    dy = torch.randn(shape, generator=g).to(dev)
    gamma = (torch.rand(C, generator=g) + 0.5).to(dev)
    beta = (torch.randn(C, generator=g) * 0.2).to(dev)
    r = torch.randn(shape, generator=g).to(dev) if res else None
    return x, dy, gamma, beta, r


def _two_kernel(ops, dy, x, y, mean, var, gamma, beta, relu, want_res, dx_out=None):
    C = x.shape[1]
    dg = torch.empty(C, device=x.device); db = torch.empty(C, device=x.device)
    s1, s2, _ = ops.bn_bwd_reduce(dy, x, y, mean, var, gamma, beta, 1e-5, relu, dgamma=dg, dbeta=db)
    cnt = float(x.numel() // C)
    dx, dres = ops.bn_bwd_apply(dy, x, y, mean, var, gamma, beta, 1e-5, s1, s2, cnt, relu, want_res, dx_out)
    return s1.clone(), s2.clone(), dg, db, dx, dres


def _fused(ops, dy, x, y, mean, var, gamma, beta, relu, want_res, dx_out=None):
    C = x.shape[1]
    dg = torch.empty(C, device=x.device); db = torch.empty(C, device=x.device)
    cnt = float(x.numel() // C)
    out = ops.bn_bwd_fused(dy, x, y, mean, var, gamma, beta, 1e-5, cnt, relu, want_res, dx_out, dgamma=dg, dbeta=db)
    assert out is not None, "fused kernel refused a supported shape"
    s1, s2, _, dx, dres = out
    return s1.clone(), s2.clone(), dg, db, dx, dres


def _same(a, b, what):
    for k, (u, v) in enumerate(zip(a, b)):
        if u is None or v is None:
            assert u is None and v is None, (what, k)
            continue
        assert torch.equal(u, v), (what, k, (u.double() - v.double()).abs().max().item())


# (N, C, H, W): one chunk; a ragged last chunk; HW not a multiple of 1024; several chunks; 256 | HW for the bit mask
SHAPES = [(2, 8, 16, 16), (3, 5, 36, 52), (2, 6, 100, 100), (4, 16, 64, 128), (1, 3, 4, 4), (5, 33, 48, 64)]


@pytest.mark.parametrize("shape", SHAPES)
def test_fused_equals_two_kernel_path(cuda, shape):
    from dcfp_amd import ops
    x, dy, gamma, beta, r = _inputs(shape, cuda, 11 + shape[1])
    mean, var = ops.bn_stats(x)
    N, C, H, W = shape
    # relu 0: no mask; 2: mask re-derived from x; 1: mask from the saved output of a residual BatchNorm
    y_res = ops.bn_apply(x, mean, var, gamma, beta, 1e-5, r, True)
    for relu, y, want_res in [(0, None, False), (2, None, False), (1, y_res, True), (1, y_res, False)]:
        a = _two_kernel(ops, dy, x, y, mean, var, gamma, beta, relu, want_res)
        b = _fused(ops, dy, x, y, mean, var, gamma, beta, relu, want_res)
        _same(a, b, (shape, relu, want_res))
    if (H * W) % 256 == 0:
        ym = ops.bn_apply_relu_mask(x, mean, var, gamma, beta, 1e-5, r)
        assert ym is not None
        a = _two_kernel(ops, dy, x, ym[1], mean, var, gamma, beta, 3, True)
        b = _fused(ops, dy, x, ym[1], mean, var, gamma, beta, 3, True)
        _same(a, b, (shape, 3))
    torch.cuda.synchronize()
    ops.check_fused_status()


@pytest.mark.parametrize("shape", [(2, 12, 32, 64), (2, 8, 64, 128)])     # (one chunk per channel; two full chunks)
def test_fused_pitched_dx_and_strided_dy(cuda, shape):
    from dcfp_amd import ops
    N, C, H, W = shape
    x, _, gamma, beta, _ = _inputs(shape, cuda, 5)
    g = torch.Generator().manual_seed(6)
    wide = torch.randn(N, 3 * C, H, W, generator=g).to(cuda)
    dy = wide[:, C:2 * C]                                   # a channel slice: images 3*C*H*W apart
    mean, var = ops.bn_stats(x)
    pa = ops.pitched_buffer(shape, W + 4, "t_fused_a", cuda)
    pb = ops.pitched_buffer(shape, W + 4, "t_fused_b", cuda)
    pa.zero_(); pb.zero_()
    a = _two_kernel(ops, dy, x, None, mean, var, gamma, beta, 2, False, dx_out=pa)
    b = _fused(ops, dy, x, None, mean, var, gamma, beta, 2, False, dx_out=pb)
    _same(a[:4], b[:4], "sums")
    assert torch.equal(pa, pb)
    # the tails behind the rows stay zero (the kernels write the W live floats only)
    assert float(pb.as_strided((N, C, H, 4), pb.stride(), pb.storage_offset() + W).abs().max()) == 0.0
    dense = _two_kernel(ops, dy, x, None, mean, var, gamma, beta, 2, False)
    assert torch.equal(dense[4], pb)


def test_fused_model_sizes_and_repeated_calls(cuda):
    """layer3's 256-channel tensors (16 blocks per channel), the stem's 64-channel full-resolution tensor (256 blocks per
    channel) - and 40 calls in a row on the same hand-off buffer with changing inputs: every call's result equals the
    two-kernel path's (a granule left by an earlier call, or by another shape, must not be taken for this call's)."""
    from dcfp_amd import ops
    for shape in [(4, 256, 128, 256), (4, 64, 512, 1024), (4, 1024, 128, 256)]:
        x, dy, gamma, beta, _ = _inputs(shape, cuda, 3, res=False)
        mean, var = ops.bn_stats(x)
        a = _two_kernel(ops, dy, x, None, mean, var, gamma, beta, 2, False)
        b = _fused(ops, dy, x, None, mean, var, gamma, beta, 2, False)
        _same(a, b, shape)
        del x, dy, a, b
    shape = (4, 64, 64, 128)
    x, dy, gamma, beta, _ = _inputs(shape, cuda, 9, res=False)
    mean, var = ops.bn_stats(x)
    used0 = ops.FUSED_BN_USED[0]
    for it in range(40):
        dy.mul_(1.01).add_(0.001 * it)
        a = _two_kernel(ops, dy, x, None, mean, var, gamma, beta, 2, False)
        b = _fused(ops, dy, x, None, mean, var, gamma, beta, 2, False)
        _same(a, b, ("repeat", it))
    assert ops.FUSED_BN_USED[0] == used0 + 40
    torch.cuda.synchronize()
    ops.check_fused_status()


def test_autograd_path_takes_the_fused_kernel_and_matches_the_switch(cuda, monkeypatch):
    """ops.batch_norm_act's backward: fused by default, two-kernel with BN_BWD_FUSED off - identical gradients."""
    from dcfp_amd import ops
    shape = (2, 24, 40, 48)
    x, dy, gamma, beta, r = _inputs(shape, cuda, 21)

    def run():
        xg = x.clone().requires_grad_(True); gg = gamma.clone().requires_grad_(True); bg = beta.clone().requires_grad_(True)
        rg = r.clone().requires_grad_(True)
        rm, rv = torch.zeros(shape[1], device=cuda), torch.ones(shape[1], device=cuda)
        y = ops.batch_norm_act(xg, gg, bg, rm, rv, rg, True, True, 0.1, 1e-5, False)
        y.backward(dy)
        return xg.grad, gg.grad, bg.grad, rg.grad

    n0 = ops.FUSED_BN_USED[0]
    a = run()
    assert ops.FUSED_BN_USED[0] == n0 + 1
    monkeypatch.setattr(ops, "BN_BWD_FUSED", False)
    b = run()
    assert ops.FUSED_BN_USED[0] == n0 + 1
    _same(a, b, "autograd")
