"""Per-op parity of the HIP kernels (through the C-ABI) against a CPU fp32/fp64 restatement
of the ATen ops the reference invokes.  Tolerances: SURVEY.md Appendix D item 1 (per-op
~1e-5 relative, scaled for long reductions)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    a = a.double().cpu(); b = b.double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def max_err(a, b):
    return (a.double().cpu() - b.double().cpu()).abs().max().item()


CONV_CASES = [
    # N, Cin, H, W, Cout, k, stride, pad, dil
    (2, 64, 16, 24, 256, 1, 1, 0, 1),
    (2, 256, 16, 24, 64, 1, 1, 0, 1),
    (1, 1024, 8, 8, 256, 1, 1, 0, 1),
    (2, 47, 9, 13, 95, 1, 1, 0, 1),        # pruned widths, odd pixels
    (2, 256, 12, 16, 512, 1, 2, 0, 1),      # downsample 1x1 stride 2
    (2, 64, 16, 24, 64, 3, 1, 1, 1),
    (2, 256, 16, 24, 256, 3, 1, 2, 2),      # layer3 conv2
    (1, 96, 20, 28, 64, 3, 1, 4, 4),
    (1, 128, 24, 40, 160, 3, 1, 12, 12),    # ASPP-like dilation
    (2, 128, 17, 23, 128, 3, 2, 1, 1),      # layer2.0.conv2 (stride 2), odd sizes
    (2, 3, 33, 47, 64, 3, 2, 1, 1),         # stem
    (2, 47, 11, 15, 95, 3, 1, 1, 1),        # pruned widths
    (2, 256, 16, 24, 19, 1, 1, 0, 1),       # classifier (bias)
    (2, 2048, 1, 1, 256, 1, 1, 0, 1),       # ASPP image-pool conv
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_dgrad_wgrad(cuda, case):
    from dcfp_amd import ops
    N, Cin, H, W, Cout, k, s, p, d = case
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)
    b = torch.randn(Cout, generator=g) if Cout == 19 else None
    x64 = x.double().requires_grad_(True); w64 = w.double().requires_grad_(True)
    b64 = b.double().requires_grad_(True) if b is not None else None
    y64 = F.conv2d(x64, w64, b64, s, p, d)
    dy = torch.randn(y64.shape, generator=g)
    y64.backward(dy.double())
    y32 = F.conv2d(x, w, b, s, p, d)

    xg = x.to(cuda).requires_grad_(True); wg = w.to(cuda).requires_grad_(True)
    bg = b.to(cuda).requires_grad_(True) if b is not None else None
    yg = ops.conv2d(xg, wg, bg, s, p, d)
    assert tuple(yg.shape) == tuple(y64.shape)
    yg.backward(dy.to(cuda))
    torch.cuda.synchronize()
    K = Cin * k * k
    tol = 3e-6 * max(1.0, math.sqrt(K) / 8)
    ref_noise = rel_err(y32, y64)
    assert rel_err(yg, y64) < max(tol, 3 * ref_noise), ("fwd", rel_err(yg, y64), ref_noise)
    assert rel_err(xg.grad, x64.grad) < max(tol, 1e-5), ("dgrad", rel_err(xg.grad, x64.grad))
    assert rel_err(wg.grad, w64.grad) < 2e-5, ("wgrad", rel_err(wg.grad, w64.grad))
    if b is not None:
        assert rel_err(bg.grad, b64.grad) < 1e-5


@pytest.mark.parametrize("shape,relu,res", [
    ((4, 64, 16, 24), True, False), ((2, 256, 9, 13), True, True), ((4, 47, 8, 8), False, False),
    ((2, 256, 1, 1), True, False), ((3, 128, 33, 31), False, True),
])
def test_bn_act_fwd_bwd(cuda, shape, relu, res):
    from dcfp_amd import ops
    g = torch.Generator().manual_seed(7)
    N, C, H, W = shape
    x = torch.randn(shape, generator=g) * 2 + 0.5
    gamma = torch.rand(C, generator=g) + 0.5
    beta = torch.randn(C, generator=g) * 0.1
    r = torch.randn(shape, generator=g) if res else None
    dy = torch.randn(shape, generator=g)
    rm, rv = torch.zeros(C), torch.ones(C)

    def ref(dt):
        xx = x.to(dt).requires_grad_(True); gg = gamma.to(dt).requires_grad_(True)
        bb = beta.to(dt).requires_grad_(True)
        rr = r.to(dt).requires_grad_(True) if res else None
        rm_, rv_ = rm.to(dt).clone(), rv.to(dt).clone()
        y = F.batch_norm(xx, rm_, rv_, gg, bb, True, 0.1, 1e-5)
        if res:
            y = y + rr
        if relu:
            y = F.relu(y)
        y.backward(dy.to(dt))
        return y, xx.grad, gg.grad, bb.grad, (rr.grad if res else None), rm_, rv_

    y64, dx64, dg64, db64, dr64, rm64, rv64 = ref(torch.float64)
    xg = x.to(cuda).requires_grad_(True); gg = gamma.to(cuda).requires_grad_(True)
    bg = beta.to(cuda).requires_grad_(True)
    rg = r.to(cuda).requires_grad_(True) if res else None
    rmg, rvg = rm.to(cuda), rv.to(cuda)
    yg = ops.batch_norm_act(xg, gg, bg, rmg, rvg, rg, relu, True, 0.1, 1e-5, False)
    yg.backward(dy.to(cuda))
    torch.cuda.synchronize()
    # ReLU masks may flip for |pre-activation| ~ 1e-7: compare with a small absolute slack
    assert max_err(yg, y64) < 2e-5
    assert max_err(rmg, rm64) < 1e-6 and max_err(rvg, rv64) < 1e-5
    assert rel_err(dg64, gg.grad) < 1e-4 and rel_err(db64, bg.grad) < 1e-4
    assert rel_err(dx64, xg.grad) < 1e-4
    if res:
        assert rel_err(dr64, rg.grad) < 1e-5


@pytest.mark.parametrize("shape", [(2, 16, 17, 23), (1, 128, 32, 64), (2, 8, 5, 5)])
def test_maxpool(cuda, shape):
    from dcfp_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(shape, generator=g)
    x[0, 0, 0, :3] = float("-inf")
    xr = x.clone().requires_grad_(True)
    y = F.max_pool2d(xr, 3, 2, 1)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    xg = x.to(cuda).requires_grad_(True)
    yg = ops.maxpool3x3s2(xg)
    yg.backward(dy.to(cuda))
    assert torch.equal(yg.cpu(), y.detach())
    assert max_err(xg.grad, xr.grad) < 1e-6


def test_global_pool_broadcast(cuda):
    from dcfp_amd import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 64, 9, 13, generator=g)
    xr = x.clone().requires_grad_(True)
    v = F.adaptive_avg_pool2d(xr, 1)
    up = F.interpolate(v, size=(9, 13), mode="bilinear", align_corners=True)
    dy = torch.randn(up.shape, generator=g)
    up.backward(dy)
    xg = x.to(cuda).requires_grad_(True)
    vg = ops.global_avg_pool(xg)
    upg = ops.broadcast_to_hw(vg, 9, 13)
    upg.backward(dy.to(cuda))
    assert max_err(vg, v) < 1e-6 and max_err(upg, up) < 1e-6
    assert max_err(xg.grad, xr.grad) < 1e-6


@pytest.mark.parametrize("align", [True, False])
@pytest.mark.parametrize("hw,HW", [((9, 13), (65, 97)), ((8, 8), (64, 64)), ((5, 7), (5, 7)), ((3, 3), (1, 1))])
def test_upsample_bilinear(cuda, align, hw, HW):
    from dcfp_amd import ops
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 5, *hw, generator=g)
    xr = x.clone().requires_grad_(True)
    y = F.interpolate(xr, size=HW, mode="bilinear", align_corners=align)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    xg = x.to(cuda).requires_grad_(True)
    yg = ops.upsample_bilinear(xg, HW, align)
    yg.backward(dy.to(cuda))
    assert max_err(yg, y) < 2e-6
    assert max_err(xg.grad, xr.grad) < 2e-5


@pytest.mark.parametrize("align", [True, False])
@pytest.mark.parametrize("hw,HW,C", [((9, 13), (65, 97), 19), ((8, 16), (64, 128), 19), ((5, 5), (33, 33), 7)])
def test_upsample_ce(cuda, align, hw, HW, C):
    from dcfp_amd import ops
    g = torch.Generator().manual_seed(13)
    N = 2
    z = torch.randn(N, C, *hw, generator=g) * 3
    lab = torch.randint(0, C, (N, *HW), generator=g)
    lab[torch.rand(N, *HW, generator=g) < 0.1] = 255
    lab[1, :4] = 255
    zr = z.double().requires_grad_(True)
    up = F.interpolate(zr, size=HW, mode="bilinear", align_corners=align)
    loss = F.cross_entropy(up, lab, ignore_index=255)
    loss.backward()
    zg = z.to(cuda).requires_grad_(True)
    lg = ops.upsample_cross_entropy(zg, lab.to(cuda), HW, align, 255)
    lg.backward()
    assert abs(lg.item() - loss.item()) < 2e-6 * max(1.0, abs(loss.item()))
    assert rel_err(zg.grad, zr.grad) < 2e-5


def test_upsample_ce_all_ignored(cuda):
    from dcfp_amd import ops
    z = torch.randn(1, 4, 3, 3)
    lab = torch.full((1, 9, 9), 255, dtype=torch.int64)
    out = ops.upsample_cross_entropy(z.to(cuda), lab.to(cuda), (9, 9), True, 255)
    assert math.isnan(out.item())      # torch: mean over zero valid pixels = nan


@pytest.mark.parametrize("case", ["kth_le", "kth_gt", "few_valid"])
def test_ohem_threshold_and_loss_vs_oracle(cuda, case):
    """OHEM on the device (zoomed ground-truth probability, k-th smallest, kept mask, CE) against
    the numpy/scipy restatement of loss/ohem.py, on the reference-pinned golden cases."""
    import os
    import numpy as np
    from oracle import ohem as oohem
    from dcfp_amd.loss.ohem import OhemCrossEntropy2d
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "ohem_threshold.npz"))
    z = torch.from_numpy(g[f"z:{case}"]); lab = torch.from_numpy(g[f"lab:{case}"]).long()
    mk = int(g[f"min_kept:{case}"])
    prob = torch.softmax(z, 1).numpy()
    new_t, th = oohem.new_target(prob, lab.numpy(), 255, 0.7, mk)
    assert float(th) == float(g[f"th:{case}"])          # oracle == reference (golden)
    zr = z.double().requires_grad_(True)
    ref_loss = F.cross_entropy(zr, torch.from_numpy(new_t).long(), ignore_index=255)
    ref_loss.backward()
    crit = OhemCrossEntropy2d(ignore_label=255, thresh=0.7, min_kept=mk)
    zg = z.to(cuda).requires_grad_(True)
    H, W = lab.shape[-2:]
    out2, lse, gtp = __import__("dcfp_amd").ops.upsample_ce_forward(zg.detach(), lab.to(cuda), (H, W), True, 255,
                                                                    want_gt_prob=True)
    th_gpu = crit.find_threshold(zg.detach(), lab.to(cuda), lse, (H, W), True)
    assert abs(th_gpu - float(th)) <= 2e-6 * max(1.0, abs(float(th))), (th_gpu, th)
    loss = crit.forward_lowres(zg, lab.to(cuda), (H, W), True)
    loss.backward()
    if math.isnan(ref_loss.item()):
        assert math.isnan(loss.item())
    else:
        assert abs(loss.item() - ref_loss.item()) < 1e-4 * max(1.0, abs(ref_loss.item()))
        assert rel_err(zg.grad, zr.grad) < 1e-3


def _ohem_select(pred, lab, thresh, min_kept, cuda, ignore=255):
    from dcfp_amd import _lib
    import ctypes as C
    p = torch.from_numpy(pred.astype("float32")).to(cuda).contiguous()
    l = torch.from_numpy(lab.astype("int32")).to(cuda).contiguous()
    out = torch.full((1,), -7.0, device=cuda)
    _lib.check(_lib.lib().dcfp_ohem_threshold_f32(C.c_void_p(p.data_ptr()), C.c_void_p(l.data_ptr()), p.numel(), ignore,
                                                  float(thresh), int(min_kept), C.c_void_p(out.data_ptr()),
                                                  C.c_void_p(torch.cuda.current_stream().cuda_stream)), "ohem_threshold")
    return out.cpu().numpy()[0]


@pytest.mark.parametrize("case", ["kth_le", "kth_gt", "few_valid"])
def test_ohem_select_kernel_bit_equal_to_reference_golden(cuda, case):
    """dcfp_ohem_threshold_f32 (device radix select) on the reference's own zoomed arrays: the threshold
    must be the golden value bit for bit (loss/ohem.py:20-48; np.partition there)."""
    import os
    import numpy as np
    import scipy.ndimage as nd
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "ohem_threshold.npz"))
    prob = torch.softmax(torch.from_numpy(g[f"z:{case}"]), 1).numpy()
    lab = g[f"lab:{case}"]
    pz = nd.zoom(prob, (1.0, 1.0, 1 / 8, 1 / 8), order=1)          # ohem.py:23-24
    lz = nd.zoom(lab, (1.0, 1 / 8, 1 / 8), order=0).astype(np.int32)
    n, c, h, w = pz.shape
    flat = np.rollaxis(pz, 1).reshape(c, -1)
    lflat = lz.ravel()
    gt = np.where(lflat != 255, flat[np.minimum(lflat, c - 1), np.arange(lflat.size)], 1.0).astype(np.float32)
    th = _ohem_select(gt, lflat, 0.7, int(g[f"min_kept:{case}"]) // 64, cuda)
    assert np.float32(th) == np.float32(g[f"th:{case}"]), (th, g[f"th:{case}"])


def test_ohem_select_kernel_vs_partition(cuda):
    """Exact k-th smallest among valid positions vs np.partition: ties, negatives, ragged n, all-ignored,
    min_kept >= valid, min_kept == 0."""
    import numpy as np
    rng = np.random.default_rng(5)
    for n in (1, 63, 1024, 4097, 131072, 300001):
        pred = rng.random(n, dtype=np.float32)
        pred[rng.random(n) < 0.2] = np.float32(0.25)            # many exact ties
        if n > 100:
            pred[:50] = -pred[:50]                              # order-preserving key handles negatives
        lab = rng.integers(0, 19, n).astype(np.int32)
        lab[rng.random(n) < 0.3] = 255
        valid = pred[lab != 255]
        for mk in (0, 1, 2, max(1, valid.size // 3), valid.size - 1, valid.size, valid.size + 5):
            if mk < 0:
                continue
            got = _ohem_select(pred, lab, 0.1, mk, cuda)
            if mk >= valid.size:
                want = np.float32(1.0)
            elif mk == 0:
                want = np.float32(0.1)
            else:
                kth = np.partition(valid, mk - 1)[mk - 1]
                want = kth if kth > np.float32(0.1) else np.float32(0.1)
            assert np.float32(got) == np.float32(want), (n, mk, got, want)
    assert _ohem_select(np.zeros(10, np.float32), np.full(10, 255, np.int32), 0.7, 0, cuda) == np.float32(1.0)


def test_ohem_fullsize_vs_oracle(cuda):
    """BASELINE config-3 geometry: 4 x 19 x 128 x 256 logits -> 1024 x 2048, min_kept 100000, against
    oracle.ohem.new_target on the materialised full-resolution softmax (loss/ohem.py:51-78).  The training
    path makes no host synchronisation and calls no ATen select (threshold stays on the device)."""
    import numpy as np
    from oracle import ohem as oohem
    from dcfp_amd import ops
    from dcfp_amd.loss.ohem import OhemCrossEntropy2d
    N, Cc, h, w, H, W = 4, 19, 128, 256, 1024, 2048
    g = torch.Generator().manual_seed(21)
    z = torch.randn(N, Cc, h, w, generator=g) * 2.0
    lab = torch.randint(0, Cc, (N, H, W), generator=g)
    lab[torch.rand(N, H, W, generator=g) < 0.05] = 255
    # make the label class likely on most pixels so that probabilities straddle 0.7
    zl = F.interpolate(F.one_hot(lab.clamp(max=Cc - 1), Cc).permute(0, 3, 1, 2).float(), size=(h, w), mode="bilinear",
                       align_corners=True)
    z = z + 4.0 * zl * torch.rand(N, 1, h, w, generator=g)
    for mk, thresh in ((100000, 0.7), (100000 * 40, 0.3)):        # threshold = thresh, and threshold = k-th value
        up = F.interpolate(z, size=(H, W), mode="bilinear", align_corners=True)
        prob = torch.softmax(up, 1).numpy()
        new_t, th = oohem.new_target(prob, lab.numpy(), 255, thresh, mk)
        del prob
        crit = OhemCrossEntropy2d(ignore_label=255, thresh=thresh, min_kept=mk)
        zg = z.to(cuda).requires_grad_(True); lg = lab.to(cuda)
        out2, lse, gtp = ops.upsample_ce_forward(zg.detach(), lg, (H, W), True, 255, want_gt_prob=True)
        thr = crit.threshold_device(zg.detach(), lg, lse, (H, W), True)
        assert thr.is_cuda and thr.numel() == 1
        assert abs(thr.item() - float(th)) <= 2e-6 * max(1.0, abs(float(th))), (thr.item(), th)
        loss = crit.forward_lowres(zg, lg, (H, W), True)
        loss.backward()
        ref = F.cross_entropy(up.double(), torch.from_numpy(new_t).long(), ignore_index=255)
        kept_ref = int((new_t != 255).sum())
        kept = int(((gtp <= thr) & (lg != 255)).sum().item())
        assert abs(kept - kept_ref) <= max(8, 2e-5 * kept_ref), (kept, kept_ref)     # pixels within an ulp of the threshold
        assert abs(loss.item() - ref.item()) < 1e-4 * max(1.0, abs(ref.item())), (loss.item(), ref.item())
        assert torch.isfinite(zg.grad).all()


@pytest.mark.parametrize("loss_type", ["ce", "ohem"])
def test_class_weighted_criterions(cuda, loss_type):
    """balance_weight=True: nn.CrossEntropyLoss(weight=dataset.class_weights, ignore_index) in CriterionDSN
    (loss/criterion.py:54-60) and in CriterionOhemDSN (loss/ohem.py:105-110), both heads, against the same
    torch losses on the materialised upsampled logits (OHEM relabelling by the oracle)."""
    import numpy as np
    from oracle import ohem as oohem
    from dcfp_amd.loss.criterion import build_criterions
    N, Cc, h, w, H, W = 2, 7, 9, 13, 65, 97

    class DS:
        ignore_label = 255; num_classes = Cc
        class_weights = torch.tensor([0.8, 1.3, 0.5, 2.0, 1.0, 0.25, 1.7])
    g = torch.Generator().manual_seed(9)
    z0 = (torch.randn(N, Cc, h, w, generator=g) * 2).requires_grad_(True)
    z1 = torch.randn(N, Cc, h, w, generator=g).requires_grad_(True)
    lab = torch.randint(0, Cc, (N, H, W), generator=g)
    lab[torch.rand(N, H, W, generator=g) < 0.1] = 255
    para = {"ds_weight": 0.4, "balance_weight": True}
    if loss_type == "ohem":
        para.update(ohem_thres=0.3, ohem_keep=64 * 40)
    crit = build_criterions(loss_type, DS(), para)
    up0 = F.interpolate(z0.double(), size=(H, W), mode="bilinear", align_corners=True)
    up1 = F.interpolate(z1.double(), size=(H, W), mode="bilinear", align_corners=True)
    t0 = lab
    if loss_type == "ohem":
        nt, _ = oohem.new_target(torch.softmax(up0.detach().float(), 1).numpy(), lab.numpy(), 255, 0.3, 64 * 40)
        t0 = torch.from_numpy(nt).long()
    wd = DS.class_weights.double()
    ref = F.cross_entropy(up0, t0, weight=wd, ignore_index=255) + 0.4 * F.cross_entropy(up1, lab, weight=wd, ignore_index=255)
    ref.backward()
    g0 = z0.detach().to(cuda).requires_grad_(True); g1 = z1.detach().to(cuda).requires_grad_(True)
    loss = crit.forward_lowres([g0, g1], lab.to(cuda), (H, W), True)["loss"]
    loss.backward()
    assert abs(loss.item() - ref.item()) < 2e-5 * max(1.0, abs(ref.item())), (loss.item(), ref.item())
    assert rel_err(g0.grad, z0.grad) < 1e-4 and rel_err(g1.grad, z1.grad) < 1e-4


@pytest.mark.parametrize("tag", ["a", "b"])
def test_gsrl_loss_vs_reference_golden(cuda, tag):
    """CriterionGsrlDSN through the fused HIP kernels vs the reference's output (golden)."""
    import os
    import numpy as np
    from dcfp_amd.loss.criterion import build_criterions

    class DS:
        ignore_label = 255; num_classes = 19; class_weights = None
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "gsrl.npz"))
    H, W, align = [int(v) for v in g[f"meta:{tag}"]]
    crit = build_criterions("gsrl", DS(), {"ds_weight": 0.4})
    z0 = torch.from_numpy(g[f"z0:{tag}"]).to(cuda).requires_grad_(True)
    z1 = torch.from_numpy(g[f"z1:{tag}"]).to(cuda).requires_grad_(True)
    labels = {"ori": torch.from_numpy(g[f"lab:{tag}"]).to(cuda), "weight": torch.from_numpy(g[f"wgt:{tag}"]).to(cuda)}
    loss = crit.forward_lowres([z0, z1], labels, (H, W), bool(align))["loss"]
    loss.backward()
    assert abs(loss.item() - float(g[f"loss:{tag}"])) < 2e-6 * max(1.0, abs(float(g[f"loss:{tag}"])))
    assert rel_err(z0.grad, torch.from_numpy(g[f"g0:{tag}"])) < 2e-5
    assert rel_err(z1.grad, torch.from_numpy(g[f"g1:{tag}"])) < 2e-5


@pytest.mark.parametrize("world,C", [(1, 64), (3, 257), (8, 19)])
def test_syncbn_combine_kernel(cuda, world, C):
    """Pooled SyncBatchNorm statistics from the all-gathered per-rank rows (mean, var, count) —
    the device-side half of ops.sync_bn_stats — against the parallel-variance formula in fp64."""
    import ctypes as C_
    from dcfp_amd import _lib, ops
    g = torch.Generator().manual_seed(5)
    means = torch.randn(world, C, generator=g)
    vars_ = torch.rand(world, C, generator=g) + 0.1
    counts = torch.tensor([float(1000 + 37 * r) for r in range(world)])
    rows = torch.cat([means, vars_, counts[:, None]], dim=1).contiguous()
    tot = counts.double().sum()
    gm = (means.double() * counts.double()[:, None]).sum(0) / tot
    gv = ((vars_.double() + (means.double() - gm) ** 2) * counts.double()[:, None]).sum(0) / tot
    dev_rows = rows.to(cuda)
    out = torch.empty(2 * C + 1, device=cuda)
    rm = torch.randn(C, generator=g).to(cuda); rv = (torch.rand(C, generator=g) + 0.5).to(cuda)
    nbt = torch.tensor(3, dtype=torch.int64, device=cuda)
    rm0, rv0 = rm.clone(), rv.clone()
    run = ops._bn_run(rm, rv, 0.1, nbt)
    rc = _lib.lib().dcfp_syncbn_combine_f32(ops._p(dev_rows), world, C, ops._p(out), ops._p(out[C:]),
                                            ops._p(out[2 * C:]), ops._rp(run), ops._stream())
    assert rc == 0
    torch.cuda.synchronize()
    assert max_err(out[:C], gm) < 1e-6
    assert max_err(out[C:2 * C], gv) < 1e-6
    assert abs(out[2 * C].item() - tot.item()) < 1e-3
    # the kernel and the host path evaluate ONE formula (ops.syncbn_combine_reference): bit-equal
    rm_, rv_, tot_ = ops.syncbn_combine_reference(dev_rows, C)
    assert torch.equal(out[:C], rm_) and torch.equal(out[C:2 * C], rv_) and torch.equal(out[2 * C:], tot_)
    # running statistics from the POOLED statistics (nn.SyncBatchNorm), counter incremented on the device
    n = tot.item()
    assert max_err(rm, 0.9 * rm0.double() + 0.1 * gm.to(cuda)) < 1e-6
    assert max_err(rv, 0.9 * rv0.double() + 0.1 * gv.to(cuda) * (n / (n - 1))) < 1e-6
    assert nbt.item() == 4


@pytest.mark.parametrize("Cout,k", [(256, 1), (128, 3), (64, 1), (512, 1), (256, 3)])
def test_conv_fused_bn_stats(cuda, Cout, k):
    """BatchNorm batch statistics emitted by the conv epilogue (Welford partials over 128-pixel
    runs, merged in fp64) against the statistics of the conv output itself and bn_stats."""
    from dcfp_amd import ops
    g = torch.Generator().manual_seed(11)
    N, Cin, H, W = 2, 24, 128, 256
    x = (torch.randn(N, Cin, H, W, generator=g) * 1.5 + 0.7).to(cuda)
    w = (torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k) + 0.05).to(cuda)
    y, stats = ops.conv2d_fwd(x, w, None, 1, k // 2, 1, want_stats=True)
    assert stats is not None, "shape should route to a tile with fused statistics"
    y_plain = ops.conv2d_fwd(x, w, None, 1, k // 2, 1)
    assert torch.equal(y, y_plain)
    m64 = y.double().mean(dim=(0, 2, 3)); v64 = y.double().var(dim=(0, 2, 3), unbiased=False)
    assert max_err(stats[0], m64) < 2e-6 * max(1.0, m64.abs().max().item())
    assert ((stats[1].double().cpu() - v64.cpu()).abs() / v64.cpu()).max().item() < 2e-6
    m2, v2 = ops.bn_stats(y)
    assert max_err(stats[0], m2) < 2e-6 and ((stats[1] - v2).abs() / v2).max().item() < 2e-6


def test_bn_relu_bitmask_path(cuda):
    """Residual BatchNorm+ReLU whose forward writes the ReLU mask as one bit per element and whose
    backward kernels read those bits (relu mode 3) — against the variant that re-reads y (mode 1)."""
    from dcfp_amd import ops
    g = torch.Generator().manual_seed(21)
    shape = (3, 40, 16, 48)                       # HW = 768 = 3 x 256
    x = (torch.randn(shape, generator=g) * 1.3).to(cuda)
    res = torch.randn(shape, generator=g).to(cuda)
    dy = torch.randn(shape, generator=g).to(cuda)
    gamma = (torch.rand(40, generator=g) + 0.5).to(cuda); beta = (torch.randn(40, generator=g) * 0.2).to(cuda)
    mean, var = ops.bn_stats(x)
    ym = ops.bn_apply_relu_mask(x, mean, var, gamma, beta, 1e-5, res)
    assert ym is not None
    y, mask = ym
    y_ref = ops.bn_apply(x, mean, var, gamma, beta, 1e-5, res, True)
    assert torch.equal(y, y_ref)
    s1a, s2a, dga = ops.bn_bwd_reduce(dy, x, y_ref, mean, var, gamma, beta, 1e-5, 1)
    s1b, s2b, dgb = ops.bn_bwd_reduce(dy, x, mask, mean, var, gamma, beta, 1e-5, 3)
    assert torch.equal(s1a, s1b)                   # same mask, same summation order
    # (the two template instantiations may contract g*(x-mean) differently: last-digit differences)
    assert rel_err(s2b, s2a) < 1e-6 and rel_err(dgb, dga) < 1e-6
    cnt = float(shape[0] * shape[2] * shape[3])
    dxa, dra = ops.bn_bwd_apply(dy, x, y_ref, mean, var, gamma, beta, 1e-5, s1a, s2a, cnt, 1, True)
    dxb, drb = ops.bn_bwd_apply(dy, x, mask, mean, var, gamma, beta, 1e-5, s1a, s2a, cnt, 3, True)
    assert torch.equal(dxa, dxb) and torch.equal(dra, drb)


@pytest.mark.parametrize("N,Cin,Cout,bias", [(4, 2048, 256, False), (10, 100, 37, True), (2, 83, 256, False)])
def test_conv1x1_on_a_1x1_map_gemv(cuda, N, Cin, Cout, bias):
    """The ASPP image-pool branch's 1x1 conv runs on an N x C x 1 x 1 tensor (aspp.py:56-61): matrix-vector kernels
    (conv_gemv.hip) for forward, dgrad (+accumulate) and weight gradient, against fp64."""
    from dcfp_amd import _lib, ops
    g = torch.Generator().manual_seed(17)
    x = torch.randn(N, Cin, 1, 1, generator=g)
    w = torch.randn(Cout, Cin, 1, 1, generator=g) / Cin ** 0.5
    b = torch.randn(Cout, generator=g) if bias else None
    dy = torch.randn(N, Cout, 1, 1, generator=g)
    desc = ops._desc(x.shape, w.shape, 1, 0, 1)
    assert ops.conv_kernel_name(desc, _lib.CONV_FWD) == "gemv_1x1_map_kernel"
    y = ops.conv2d_fwd(x.to(cuda), w.to(cuda), None if b is None else b.to(cuda), 1, 0, 1)
    dx = ops.conv2d_dgrad(dy.to(cuda), w.to(cuda), tuple(x.shape), 1, 0, 1)
    seed = torch.randn(x.shape, generator=g).to(cuda)
    acc = seed.clone()
    ops.conv2d_dgrad(dy.to(cuda), w.to(cuda), tuple(x.shape), 1, 0, 1, out=acc, accumulate=True)
    dw, db = ops.conv2d_wgrad(dy.to(cuda), x.to(cuda), tuple(w.shape), 1, 0, 1, need_bias=bias)
    torch.cuda.synchronize()
    ref = torch.nn.functional.conv2d(x.double(), w.double(), None if b is None else b.double())
    refdx = torch.einsum("nm,mc->nc", dy.double().view(N, Cout), w.double().view(Cout, Cin)).view(N, Cin, 1, 1)
    refdw = torch.einsum("nm,nc->mc", dy.double().view(N, Cout), x.double().view(N, Cin)).view(Cout, Cin, 1, 1)

    def emax(a, r):
        return float((a.cpu().double() - r).abs().max() / r.abs().max())
    assert emax(y, ref) < 3e-6 and emax(dx, refdx) < 3e-6 and emax(acc, refdx + seed.cpu().double()) < 3e-6
    assert emax(dw, refdw) < 3e-6
    if bias:
        assert emax(db, dy.double().sum((0, 2, 3))) < 3e-6
