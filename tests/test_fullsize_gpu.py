"""Parity at BASELINE.json's full sizes.

config (2) DeepLabv3-R50 2x3x512x1024: one whole iteration (forward, CE + 0.4 CE, backward, EIC)
on the MI355X against the CPU oracle on the same closed-form weights and inputs - the sizes at
which the 256 x 256 LDS-DMA conv kernels, the split-K wgrads and the multi-block BatchNorm
reductions are the ones that run (test_model_gpu.py's 65 x 65 cases never reach them).

config (3) DeepLabv3-R101 4x3x1024x2048: too large for the oracle in a test, so size-independent
properties instead: (a) the step is bit-reproducible (every reduction has a fixed order),
(b) convolution is exactly homogeneous under a power-of-two scale (x -> 2x doubles every output
bit for bit in fp32) on the model's FLOP-dominant shapes, (c) accumulate-dgrad == plain dgrad +
seed, (d) the per-rank mean-of-valid-pixels loss is invariant to relabelling ignored pixels.
Tolerances: SURVEY.md Appendix D (loss 1e-5 relative, gradients rel-L2 5e-2, logits 1e-3)."""
import os

import numpy as np
import pytest
import torch

from oracle import fill, model as omodel
from oracle.train_step import CpuTrainer
from _parity import check_per_tensor

pytestmark = pytest.mark.gpu
WINO = os.environ.get("DCFP_CONV_WINOGRAD", "1") != "0"
BB = {"os": 8, "mg_unit": [1, 2, 4], "inplanes": 128, "pretrained": False}


class _DS:
    ignore_label = 255
    num_classes = 19
    class_weights = None


def _build(backbone, device, closed_form=True):
    from dcfp_amd import networks
    from dcfp_amd.loss.criterion import build_criterions
    crit = build_criterions("ce", _DS(), {"ds_weight": 0.4})
    m = networks.deeplabv3.Seg_Model(backbone=backbone, backbone_para=dict(BB), num_classes=19,
                                     align_corner=True, criterion=crit, deepsup=True)
    if closed_form:
        m.load_state_dict(fill.closed_form_state(m.state_dict()))
    m.conv_deepsup[3].p = 0.0
    return m.to(device).train()


def _iteration_vs_oracle(backbone, cuda, capsys):
    from dcfp_amd import pruners
    N, H, W = 2, 512, 1024
    m = _build(backbone, cuda)
    sd0 = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    x = fill.closed_form_input(N, H, W)
    lab = fill.closed_form_labels(N, H, W)
    tp = pruners.dcfp_pruning(m, 0.999)
    loss = m(x.to(cuda), lab.to(cuda), deepsup=True)["loss"]
    loss.backward()
    tp.step(m)
    torch.cuda.synchronize()

    cfg = omodel.Cfg(model="deeplabv3", backbone=backbone, align_corner=True)
    cpu = CpuTrainer(sd0, cfg, r=0.999)
    closs, outs, lowres = cpu.step(x, lab, update=False)
    assert abs(loss.item() - closs) <= 2e-5 * max(1.0, abs(closs)), (loss.item(), closs)

    # acceptance as SURVEY.md App. D item 1 states it, PER TENSOR: error against an fp64 run of the oracle, bounded by
    # max(5e-2, 3x the fp32 oracle's own error on THAT tensor against that run) - tests/_parity.py (the only tensors held
    # to the oracle's worst tensor instead are the three in front of the N-sample BatchNorm aspp.global_avg_pool.2)
    cpu64 = CpuTrainer(sd0, cfg, r=0.999, dtype=torch.float64)
    _, outs64, _ = cpu64.step(x.double(), lab, update=False)          # (its forward also serves the logits check below)
    params = dict(m.named_parameters())
    g32, g64 = cpu.params(), cpu64.params()
    names, mine, ref = [], [], []
    for k in g64:
        b = g64[k].grad
        names.append(k)
        mine.append(((params[k].grad.double().cpu() - b).norm() / (b.norm() + 1e-30)).item())
        ref.append(((g32[k].grad.double() - b).norm() / (b.norm() + 1e-30)).item())
    dump = os.environ.get("DCFP_DUMP_GRAD_ROWS")
    check_per_tensor(mine, ref, names, f"{backbone} 2x512x1024 gradients", capsys,
                     dump and os.path.join(dump, f"grad_rows_{backbone}.json"))
    assert max(ref) < 0.2, max(ref)                  # (the oracle itself is meaningful at this size)
    # the EIC statistic the pruner consumes
    mine = torch.cat([tp.get_eic()["eic"][n].reshape(-1) for n in cpu.scored]).double().cpu().numpy()
    e32 = np.concatenate([np.asarray(cpu.eic[n], dtype=np.float64).reshape(-1) for n in cpu.scored])
    e64 = np.concatenate([np.asarray(cpu64.eic[n], dtype=np.float64).reshape(-1) for n in cpu64.scored])
    rel = np.linalg.norm(mine - e64) / np.linalg.norm(e64)
    ref_rel = np.linalg.norm(e32 - e64) / np.linalg.norm(e64)
    assert rel <= 3 * ref_rel, (rel, ref_rel)
    # logits of the same (train-mode) forward on a fresh model (the step above moved the running statistics, which a
    # train-mode forward does not read): error against the fp64 oracle forward bounded by 3x the fp32 oracle's own
    m2 = _build(backbone, cuda)
    with torch.no_grad():
        outs2 = m2(x.to(cuda), None, deepsup=True)
    for mine_o, ref32, ref64 in zip(outs2, outs, outs64):
        ref64 = ref64.detach()
        err = (mine_o.double().cpu() - ref64).abs().max().item()
        ref_err = (ref32.detach().double() - ref64).abs().max().item()
        assert err <= max(1e-3, 3 * ref_err), (err, ref_err, ref64.abs().max().item())
        rel = ((mine_o.double().cpu() - ref64).norm() / ref64.norm()).item()
        ref_rel = ((ref32.detach().double() - ref64).norm() / ref64.norm()).item()
        assert rel <= max(1e-4, 3 * ref_rel), (rel, ref_rel)


def test_config2_iteration_vs_oracle(cuda, capsys):
    _iteration_vs_oracle("resnet50", cuda, capsys)


def test_r101_iteration_vs_oracle_at_real_map_sizes(cuda, capsys):
    """DeepLabv3-R101 at 2x3x512x1024 (64 x 128 feature maps): the 23 layer3 blocks, the multi-grid layer4
    (dilation 4 / 8 / 16, networks/backbone/resnet.py:124-141) and the ASPP branches (dilation 12 / 24 / 36,
    networks/tools/aspp.py:40-47) run through the kernels the headline config uses - Winograd with its 2d x 2d
    super-blocks padded (2 * 12, 2 * 24 and 2 * 36 do not divide 64 / 128), the fused Winograd kernel on the pitched
    layer3 operands, the 256 x 256 LDS-DMA tiles - as one whole iteration against the CPU oracle: loss, every parameter
    gradient, the EIC vector, low-resolution logits against an fp64 forward (the 2 x 65 x 65 golden has 9 x 9 maps,
    where none of these engage)."""
    from dcfp_amd import ops, _lib
    if WINO:
        seen = {}
        for tag, (cin, cout, dil) in {"layer3": (256, 256, 2), "layer4": (512, 512, 4), "aspp": (2048, 256, 12)}.items():
            desc = ops._desc((2, cin, 64, 128), (cout, cin, 3, 3), 1, dil, dil)
            seen[tag] = [ops.conv_kernel_name(desc, k) for k in (_lib.CONV_FWD, _lib.CONV_DGRAD, _lib.CONV_WGRAD)]
        assert all(any(n.startswith("winograd_f2x2_3x3") for n in names) for names in seen.values()), seen
    _iteration_vs_oracle("resnet101", cuda, capsys)


@pytest.fixture(scope="module")
def r101(cuda):
    torch.manual_seed(12345)
    m = _build("resnet101", cuda, closed_form=False)
    g = torch.Generator().manual_seed(12345)
    x = torch.randn(4, 3, 1024, 2048, generator=g).to(cuda)
    lab = torch.randint(0, 19, (4, 1024, 2048), generator=g)
    lab[torch.rand(lab.shape, generator=g) < 0.05] = 255
    yield m, x, lab.to(cuda)
    del m, x
    torch.cuda.empty_cache()


def _step(m, x, lab):
    for p in m.parameters():
        p.grad = None
    loss = m(x, lab, deepsup=True)["loss"]
    loss.backward()
    torch.cuda.synchronize()
    return loss.detach().clone(), [p.grad.clone() for p in m.parameters()]


def test_config3_step_is_bit_reproducible(cuda, r101):
    m, x, lab = r101
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    l1, g1 = _step(m, x, lab)
    m.load_state_dict(sd)            # running statistics back to where the first step started
    l2, g2 = _step(m, x, lab)
    assert torch.isfinite(l1) and torch.equal(l1, l2), (l1.item(), l2.item())
    bad = [n for (n, _), a, b in zip(m.named_parameters(), g1, g2) if not torch.equal(a, b)]
    assert not bad, bad[:5]
    assert all(torch.isfinite(g).all() for g in g1)


def test_config3_loss_ignores_ignored_pixels(cuda, r101):
    """CE is a mean over the rank's own valid pixels (loss/criterion.py:60): what the image holds
    under an ignored label cannot matter, and ignoring MORE pixels changes the divisor."""
    m, x, lab = r101
    with torch.no_grad():
        a = m(x, lab, deepsup=True)["loss"]
        lab2 = lab.clone(); lab2[:, ::2, :] = 255
        b = m(x, lab2, deepsup=True)["loss"]
        lab3 = lab.clone(); lab3[:, 1::2, :] = 255
        c = m(x, lab3, deepsup=True)["loss"]
        n2 = (lab2 != 255).sum().double(); n3 = (lab3 != 255).sum().double()
        # the two halves partition the valid pixels: weighted mean of the half losses == full loss
        mix = (b.double() * n2 + c.double() * n3) / (n2 + n3)
    assert abs(mix.item() - a.item()) <= 1e-5 * abs(a.item()), (a.item(), mix.item())
    assert abs(b.item() - a.item()) > 0 or abs(c.item() - a.item()) > 0


FULL_SHAPES = [
    # N, Cin, H, W, Cout, k, stride, pad, dil      
    (4, 256, 128, 256, 256, 3, 1, 2, 2),      # layer3 conv2
    (4, 256, 128, 256, 1024, 1, 1, 0, 1),     # layer3 conv3
    (4, 1024, 128, 256, 256, 1, 1, 0, 1),     # layer3 conv1
    (4, 2048, 128, 256, 256, 3, 1, 12, 12),   # ASPP
    (4, 64, 512, 1024, 128, 3, 1, 1, 1),      # stem
    (4, 128, 256, 512, 128, 3, 2, 1, 1),      # layer2.0 conv2 (stride 2)
    (4, 3, 1024, 2048, 64, 3, 2, 1, 1),       # first conv
    (4, 512, 128, 256, 19, 1, 1, 0, 1),       # classifier
]


@pytest.mark.parametrize("shape", FULL_SHAPES)
def test_config3_conv_power_of_two_homogeneity(cuda, shape):
    from dcfp_amd import ops
    N, Cin, H, W, Cout, k, s, p, d = shape
    g = torch.Generator().manual_seed(7)
    x = torch.randn(N, Cin, H, W, generator=g).to(cuda)
    w = (torch.randn(Cout, Cin, k, k, generator=g) * (Cin * k * k) ** -0.5).to(cuda)
    y = ops.conv2d_fwd(x, w, None, s, p, d)
    assert torch.equal(ops.conv2d_fwd(x * 2, w, None, s, p, d), y * 2)
    assert torch.equal(ops.conv2d_fwd(x, w * 0.25, None, s, p, d), y * 0.25)
    dy = torch.randn(y.shape, generator=g).to(cuda)
    dx = ops.conv2d_dgrad(dy, w, tuple(x.shape), s, p, d)
    assert torch.equal(ops.conv2d_dgrad(dy * 4, w, tuple(x.shape), s, p, d), dx * 4)
    seed = torch.randn(x.shape, generator=g).to(cuda)
    acc = seed.clone()
    ops.conv2d_dgrad(dy, w, tuple(x.shape), s, p, d, out=acc, accumulate=True)
    assert torch.equal(acc, seed + dx)
    dw = ops.conv2d_wgrad(dy, x, tuple(w.shape), s, p, d)[0]
    assert torch.equal(ops.conv2d_wgrad(dy * 0.5, x * 8, tuple(w.shape), s, p, d)[0], dw * 4)
    assert torch.isfinite(y).all() and torch.isfinite(dx).all() and torch.isfinite(dw).all()
    # spot values against an fp64 dot product (64 random outputs of the forward)
    idx = torch.randint(0, y.numel(), (64,), generator=g)
    xd = torch.nn.functional.pad(x.double(), (p, p, p, p)); wd = w.double()
    Ho, Wo = y.shape[2], y.shape[3]
    for i in idx.tolist():
        n, r = divmod(i, Cout * Ho * Wo); co, r = divmod(r, Ho * Wo); oy, ox = divmod(r, Wo)
        patch = xd[n, :, oy * s: oy * s + d * (k - 1) + 1: d, ox * s: ox * s + d * (k - 1) + 1: d]
        ref = (patch * wd[co]).sum().item()
        assert abs(y.view(-1)[i].item() - ref) <= 1e-4 * max(1.0, abs(ref)), (i, y.view(-1)[i].item(), ref)


# the exact config-3 launch geometries of dgrad / wgrad (split-K factor, 8 x 32 pixel tiles at W = 256,
# stride-2 dgrad) against fp64 on a channel slice; expected kernel family per pass as a routing check
SLICE_SHAPES = [
    ((4, 256, 128, 256, 256, 3, 1, 2, 2), ("igemm2_dma_kernel<9,true>", "wgrad_dma_kernel<9,true>")),     # layer3 conv2
    ((4, 256, 128, 256, 1024, 1, 1, 0, 1), ("igemm2_dma1p_kernel", "wgrad_dma_kernel<1,false>")),  # layer3 conv3
    ((4, 1024, 128, 256, 256, 1, 1, 0, 1), ("igemm2_dma1p_kernel", "wgrad_dma_kernel<1,false>")),  # layer3 conv1
    ((4, 2048, 128, 256, 256, 3, 1, 12, 12), ("igemm2_dma_kernel<9,false>", "wgrad_dma_kernel<9,false>")),  # ASPP
    ((4, 1024, 128, 256, 512, 3, 1, 1, 1), ("igemm2_dma_kernel<9,true>", "wgrad_dma_kernel<9,true>")),    # conv_deepsup.0
    ((4, 512, 128, 256, 512, 3, 1, 16, 16), ("igemm2_dma_kernel<9,false>", "wgrad_dma_kernel<9,false>")),  # layer4.2 conv2 (mg_unit 4)
    ((4, 512, 128, 256, 512, 3, 1, 8, 8), ("igemm2_dma_kernel<9,false>", "wgrad_dma_kernel<9,false>")),    # layer4.1 conv2
    ((4, 2048, 128, 256, 256, 3, 1, 36, 36), ("igemm2_dma_kernel<9,false>", "wgrad_dma_kernel<9,false>")),  # ASPP d36: dgrad stays direct with Winograd on (dead kernel rows)
    ((4, 64, 512, 1024, 128, 3, 1, 1, 1), (None, None)),                                                  # stem (dense operands here: direct kernels)
    ((4, 128, 256, 512, 128, 3, 2, 1, 1), (None, None)),                                                  # layer2.0 conv2 (stride 2)
    ((4, 256, 256, 512, 512, 1, 2, 0, 1), (None, None)),                                                  # layer2.0 downsample (stride 2)
]


@pytest.mark.parametrize("shape,kernels", SLICE_SHAPES)
def test_config3_dgrad_wgrad_vs_fp64_slice(cuda, shape, kernels):
    import math
    import torch.nn.functional as F
    from dcfp_amd import ops, _lib
    N, Cin, H, W, Cout, k, s, p, d = shape
    g = torch.Generator().manual_seed(11)
    x = torch.randn(N, Cin, H, W, generator=g)
    x = (torch.relu(x) + 0.05 * x).to(cuda)                    # post-ReLU-like statistics
    w = (torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)).to(cuda)
    desc = ops._desc(x.shape, w.shape, s, p, d)
    for want, which in zip(kernels, (_lib.CONV_DGRAD, _lib.CONV_WGRAD)):
        if want is not None:
            name = ops.conv_kernel_name(desc, which)      # (Winograd unless DCFP_CONV_WINOGRAD=0: test_winograd_gpu.py)
            assert name == want or (WINO and name.startswith("winograd_f2x2_3x3")), (name, want)
    Ho, Wo = desc.Hout, desc.Wout
    dy = (torch.randn(N, Cout, Ho, Wo, generator=g) * 1e-2).to(cuda)
    dx = ops.conv2d_dgrad(dy, w, tuple(x.shape), s, p, d)
    seed = torch.randn(x.shape, generator=g).to(cuda)
    dxa = seed.clone()
    ops.conv2d_dgrad(dy, w, tuple(x.shape), s, p, d, out=dxa, accumulate=True)
    dw = ops.conv2d_wgrad(dy, x, tuple(w.shape), s, p, d)[0]
    torch.cuda.synchronize()
    # fp64 reference for a slice of input channels: dx[:, ci] and dw[:, ci] need every output channel of
    # dy but only w[:, ci] / x[:, ci] (last channels: includes a ragged last tile where there is one)
    nci = min(Cin, 16)
    ci = slice(Cin - nci, Cin)
    try:
        x64 = x[:, ci].double().requires_grad_(True)
        w64 = w[:, ci].double().requires_grad_(True)
        F.conv2d(x64, w64, None, s, p, d).backward(dy.double())
        rdx, rdw = x64.grad, w64.grad
    except RuntimeError:                                        # no fp64 conv on the device: CPU
        x64 = x[:, ci].double().cpu().requires_grad_(True)
        w64 = w[:, ci].double().cpu().requires_grad_(True)
        F.conv2d(x64, w64, None, s, p, d).backward(dy.double().cpu())
        rdx, rdw = x64.grad.to(cuda), w64.grad.to(cuda)

    def rel(a, b):
        return ((a.double() - b).norm() / b.norm()).item()
    Kd = Cout * k * k                                           # dgrad reduction length
    tol = 3e-6 * max(1.0, math.sqrt(Kd) / 8)
    assert rel(dx[:, ci], rdx) < max(tol, 1e-5), rel(dx[:, ci], rdx)
    assert rel(dxa[:, ci], rdx + seed[:, ci].double()) < max(tol, 1e-5)
    assert rel(dw[:, ci], rdw) < 2e-5, rel(dw[:, ci], rdw)      # N*Ho*Wo-long reduction, split-K in a fixed order


# the config-3 launch geometries of the FUSED Winograd weight gradient (conv_winograd3.hip: split-K factor, blocks per XCD, the
# tile table of a 4-image batch) on the operands the model hands it - row-pitched for dilation 1 / 2, dense for >= 4 - against
# fp64 on a slice of input channels, with the routing asserted (WINO: DCFP_CONV_WINOGRAD != 0)
WGF_SHAPES = [(4, 256, 128, 256, 256, 2),      # layer3 conv2 (x 23)
              (4, 512, 128, 256, 512, 4),      # layer4.0 conv2
              (4, 2048, 128, 256, 256, 12),    # ASPP d12: 16 % padded tiles
              (4, 2048, 128, 256, 256, 36),    # ASPP d36: 27 %
              (4, 512, 128, 256, 256, 1),      # last_conv.0
              (4, 64, 512, 1024, 64, 1),       # stem conv1.3: ONE 64 x 64 block, 256 splits
              (4, 64, 512, 1024, 128, 1),      # stem conv1.6
              (4, 128, 128, 256, 128, 1)]      # layer2 conv2


@pytest.mark.parametrize("shape", WGF_SHAPES)
def test_config3_fused_winograd_wgrad_vs_fp64_slice(cuda, shape):
    import math
    import torch.nn.functional as F
    from dcfp_amd import ops, _lib
    if not WINO or os.environ.get("DCFP_WINO_WGRAD_FUSED", "1") == "0":
        pytest.skip("Winograd / fused weight gradient switched off")
    N, Cin, H, W, Cout, d = shape
    g = torch.Generator().manual_seed(19)
    x = torch.randn(N, Cin, H, W, generator=g)
    x = (torch.relu(x) + 0.05 * x).to(cuda)
    dy = (torch.randn(N, Cout, H, W, generator=g) * 1e-2).to(cuda)
    wshape = (Cout, Cin, 3, 3)
    pitch = ops.conv_pitch(tuple(x.shape), wshape, 1, d, d)
    assert (pitch > 0) == (d <= 2)
    xs, dys = x, dy
    if pitch:
        xs = ops.new_pitched(tuple(x.shape), pitch, cuda); xs.copy_(x)
        dys = ops.new_pitched(tuple(dy.shape), pitch, cuda); dys.copy_(dy)
    desc = ops._desc(x.shape, wshape, 1, d, d, pitch, pitch)
    name = ops.conv_kernel_name(desc, _lib.CONV_WGRAD)
    assert name.startswith("winograd_f2x2_3x3 wgrad fused"), name
    assert _lib.lib().dcfp_conv2d_xform_bytes(ops.C.byref(desc)) == 0        # nothing kept from the forward pass for it
    dw = ops.conv2d_wgrad(dys, xs, wshape, 1, d, d)[0]
    torch.cuda.synchronize()
    nci = min(Cin, 8)
    ci = slice(Cin - nci, Cin)
    w64 = torch.zeros(Cout, nci, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x[:, ci].double().cpu(), w64, None, 1, d, d).backward(dy.double().cpu())
    ref = w64.grad.to(cuda)
    rel = ((dw[:, ci].double() - ref).norm() / ref.norm()).item()
    assert rel < 2e-5, rel          # (the stated tolerance of the conv tests for weight gradients; measured 0.7e-6 ... 4e-6)


# ---------------------------------------------------------------- config 5 at full size: the slim R101
SLIM_CFG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "data",
                        "channel_cfg_r101_p60_synthetic.pth")


def test_config5_slim_r101_step_is_bit_reproducible(cuda):
    """The slim DeepLabv3-R101 of tools/pipeline_cfg5.sh (60 % of the FLOPs pruned, channel counts off every tile grid)
    at 4x3x1024x2048: one fine-tune step (train.py:200-205 + :255-270) twice from the same state gives the same bits -
    ragged-M kernels, Winograd on ragged widths and their split-K reductions all have a fixed order."""
    from dcfp_amd import networks, pruners
    from dcfp_amd.loss.criterion import build_criterions
    torch.manual_seed(12345)
    m = networks.deeplabv3.Seg_Model(backbone="resnet101", backbone_para=dict(BB), num_classes=19, align_corner=True,
                                     criterion=build_criterions("ce", _DS(), {"ds_weight": 0.4}), deepsup=True)
    pruners.init_pruned_model(m, torch.load(SLIM_CFG, weights_only=False))
    m.conv_deepsup[3].p = 0.0
    m = m.to(cuda).train()
    widths = sorted({mod.out_channels for mod in m.modules() if isinstance(mod, torch.nn.Conv2d)})
    assert any(wd % 32 for wd in widths), widths              # really ragged
    g = torch.Generator().manual_seed(12345)
    x = torch.randn(4, 3, 1024, 2048, generator=g).to(cuda)
    lab = torch.randint(0, 19, (4, 1024, 2048), generator=g)
    lab[torch.rand(lab.shape, generator=g) < 0.05] = 255
    lab = lab.to(cuda)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    l1, g1 = _step(m, x, lab)
    m.load_state_dict(sd)
    l2, g2 = _step(m, x, lab)
    assert torch.isfinite(l1) and torch.equal(l1, l2), (l1.item(), l2.item())
    bad = [n for (n, _), a, b in zip(m.named_parameters(), g1, g2) if not torch.equal(a, b)]
    assert not bad, bad[:5]
    assert all(torch.isfinite(gr).all() for gr in g1)
    del m, x, g1, g2
    torch.cuda.empty_cache()


# ragged launch geometries of that slim model (tools/conv_bench.py p_*), all three passes against fp64 on a channel slice
SLIM_SHAPES = [(4, 236, 128, 256, 232, 3, 1, 2, 2), (4, 204, 128, 256, 188, 3, 1, 2, 2), (4, 2048, 128, 256, 83, 3, 1, 36, 36),
               (4, 1024, 128, 256, 236, 1, 1, 0, 1)]


@pytest.mark.parametrize("shape", SLIM_SHAPES)
def test_config5_ragged_shapes_vs_fp64_slice(cuda, shape):
    import math
    import torch.nn.functional as F
    from dcfp_amd import ops
    N, Cin, H, W, Cout, k, s, p, d = shape
    g = torch.Generator().manual_seed(13)
    x = torch.randn(N, Cin, H, W, generator=g)
    x = (torch.relu(x) + 0.05 * x).to(cuda)
    w = (torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)).to(cuda)
    dy = (torch.randn(N, Cout, H, W, generator=g) * 1e-2).to(cuda)
    pitch = ops.conv_pitch(tuple(x.shape), tuple(w.shape), s, p, d)
    xs, dys = x, dy
    if pitch:                     # the model hands these convs row-pitched operands
        xs = ops.new_pitched(tuple(x.shape), pitch, cuda); xs.copy_(x)
        dys = ops.new_pitched(tuple(dy.shape), pitch, cuda); dys.copy_(dy)
    y = ops.conv2d_fwd(xs, w, None, s, p, d)
    dx = ops.conv2d_dgrad(dys, w, tuple(x.shape), s, p, d)
    dw = ops.conv2d_wgrad(dys, xs, tuple(w.shape), s, p, d)[0]
    torch.cuda.synchronize()

    def rel(a, b):
        return ((a.double() - b).norm() / b.norm()).item()
    # forward: the LAST output channels (ragged tile) need every input channel
    co = slice(Cout - min(Cout, 8), Cout)
    ry = F.conv2d(x.double().cpu(), w[co].double().cpu(), None, s, p, d)
    tol = 3e-6 * max(1.0, math.sqrt(Cin * k * k) / 8)
    assert rel(y[:, co].cpu(), ry) < max(tol, 1e-5), rel(y[:, co].cpu(), ry)
    # dgrad / wgrad: the last input channels need every output channel
    ci = slice(Cin - min(Cin, 8), Cin)
    x64 = x[:, ci].double().cpu().requires_grad_(True)
    w64 = w[:, ci].double().cpu().requires_grad_(True)
    F.conv2d(x64, w64, None, s, p, d).backward(dy.double().cpu())
    assert rel(dx[:, ci].cpu(), x64.grad) < max(3e-6 * max(1.0, math.sqrt(Cout * k * k) / 8), 1e-5)
    assert rel(dw[:, ci].cpu(), w64.grad) < 2e-5
