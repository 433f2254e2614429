"""Whole-model parity on the MI355X: the product Seg_Model (HIP kernels through the C-ABI)
against the CPU oracle on the same closed-form weights/inputs, and against the golden
vectors generated from the real reference.  Tolerances from SURVEY.md Appendix D item 1:
logits max|d| <= 1e-3, loss |d| <= 1e-5 (scaled), per-tensor gradient rel-L2 <= 5e-2 or
<= 3x the reference's own fp32-vs-fp64 error."""
import os

import numpy as np
import pytest
import torch

from oracle import fill, model as omodel, scoring
from oracle.train_step import CpuTrainer
from _parity import check_per_tensor, check_rankwise

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
BB = {"os": 8, "mg_unit": [1, 2, 4], "inplanes": 128, "pretrained": False}


class _DS:
    ignore_label = 255
    num_classes = 19
    class_weights = None


def build(model_name, backbone, align, device):
    from dcfp_amd import networks
    from dcfp_amd.loss.criterion import build_criterions
    crit = build_criterions("ce", _DS(), {"ds_weight": 0.4})
    m = getattr(networks, model_name).Seg_Model(backbone=backbone, backbone_para=dict(BB), num_classes=19,
                                                align_corner=align, criterion=crit, deepsup=True)
    m.load_state_dict(fill.closed_form_state(m.state_dict()))
    m.conv_deepsup[3].p = 0.0
    return m.to(device).train()


CASES = [("simple_r50_4x64x64", "simple", "resnet50"), ("v3_r50_2x65x65", "deeplabv3", "resnet50"),
         ("v3_r101_2x65x65", "deeplabv3", "resnet101")]


@pytest.mark.parametrize("tag,model_name,backbone", CASES)
def test_forward_backward_vs_reference_golden(cuda, capsys, tag, model_name, backbone):
    g = np.load(os.path.join(G, f"model_{tag}.npz"))
    N, H, W, align = [int(v) for v in g["meta"]]
    m = build(model_name, backbone, bool(align), cuda)
    x = fill.closed_form_input(N, H, W).to(cuda)
    lab = fill.closed_form_labels(N, H, W).to(cuda)
    out = m(x, lab, deepsup=True)
    loss = out["loss"]
    loss.backward()
    torch.cuda.synchronize()

    ref64 = float(g["loss64"]); ref32 = float(g["loss32"])
    assert abs(loss.item() - ref64) <= max(1e-5 * abs(ref64), 3 * abs(ref32 - ref64)), (loss.item(), ref32, ref64)

    # logits (inference-style call on a fresh model so BN running stats match the golden's)
    m2 = build(model_name, backbone, bool(align), cuda)
    with torch.no_grad():
        outs = m2(x, None, deepsup=True)
    for o, key, dkey in ((outs[0], "logits32", "logits_d64m32"), (outs[1], "logits_ds32", "logits_ds_d64m32")):
        l64 = g[key].astype(np.float64) + g[dkey]
        err = np.abs(o[:, :, ::2, ::2].double().cpu().numpy() - l64).max()
        ref_err = np.abs(g[dkey]).max()
        assert err <= max(1e-3, 3 * ref_err), (key, err, ref_err)

    # BN gamma / beta gradients: the statistic that feeds the EIC score
    names = g["bn_names"].tolist()
    mods = dict(m.named_modules())
    for what, attr in (("bn_wgrad", "weight"), ("bn_bgrad", "bias")):
        mine = torch.cat([getattr(mods[n], attr).grad.reshape(-1) for n in names]).double().cpu().numpy()
        r64, r32 = g[what + "64"], g[what + "32"]
        rel = np.linalg.norm(mine - r64) / np.linalg.norm(r64)
        ref_rel = np.linalg.norm(r32 - r64) / np.linalg.norm(r64)
        # bounded by the reference's own fp32-vs-fp64 noise only (no fixed 5e-2 floor)
        assert rel <= max(1e-3, 3 * ref_rel), (what, rel, ref_rel)

    # every parameter gradient through its L2 norm (fixture holds norms for all ~160-310 tensors)
    pn = g["param_names"].tolist()
    params = dict(m.named_parameters())
    mine = np.array([float(params[k].grad.double().norm()) for k in pn])
    l64 = g["grad_l2:64"]
    rel = np.abs(mine - l64) / (np.abs(l64) + 1e-12)
    # the reference's own fp32-vs-fp64 error of a tensor: the largest over its five fp32 summation orders (8 / 4 / 2 / 1
    # threads, oneDNN off - oracle/make_golden.py); one fp32 run is a single draw of that error
    variants = [str(v) for v in g["fp32_variants"]]
    ref_rel = np.max([np.abs(g["grad_l2:" + v] - l64) for v in variants], axis=0) / (np.abs(l64) + 1e-12)
    # PER TENSOR (tests/_parity.py): a tensor passes iff its error is within max(floor, 3x the reference's own fp32-vs-fp64
    # error on that tensor); floor = min(5e-2, 3x the reference's worst tensor) - 1.5e-2 on `simple`, 5e-2 on v3
    check_per_tensor(rel, ref_rel, pn, f"{tag} gradient norms", capsys)
    # ... and through a fixed-cosine projection, which (unlike a norm) sees permuted / transposed gradients:
    # a random error of relative size e moves the projection by ~ e * |g| / sqrt(2)
    proj = np.array([float((params[k].grad.double().reshape(-1) *
                            torch.cos(0.37 * torch.arange(params[k].numel(), dtype=torch.float64, device=cuda))).sum())
                     for k in pn])
    p64 = g["grad_proj:64"]
    perr = np.abs(proj - p64) / (np.abs(l64) + 1e-12)
    pref = np.max([np.abs(g["grad_proj:" + v] - p64) for v in variants], axis=0) / (np.abs(l64) + 1e-12)
    check_rankwise(perr, pref, pn, f"{tag} gradient projections", capsys)     # (why rank-wise: tests/_parity.py)
    for key in ("backbone.conv1.0", "backbone.layer1.0.conv1", "backbone.layer2.0.conv2", "last_conv.6"):
        a = params[key + ".weight"].grad.double().cpu().numpy(); b = g[f"wgrad:{key}:64"]
        rel = np.linalg.norm(a - b) / np.linalg.norm(b)
        ref_rel = np.linalg.norm(g[f"wgrad:{key}:32"] - b) / np.linalg.norm(b)
        assert rel <= max(1e-3, 3 * ref_rel), (key, rel, ref_rel)
    sd = m.state_dict()
    assert np.abs(sd["backbone.bn1.running_mean"].cpu().numpy() - g["rm:backbone.bn1:64"]).max() < 1e-5
    assert np.abs(sd["backbone.bn1.running_var"].cpu().numpy() - g["rv:backbone.bn1:64"]).max() < 1e-5


def test_eic_kernel_bit_exact_vs_golden(cuda):
    """dcfp_pruning.step through the HIP kernel on the reference's recorded (gamma, grad)."""
    g = np.load(os.path.join(G, "eic_trajectory.npz"))
    from dcfp_amd import pruners

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.bn_a = torch.nn.BatchNorm2d(37)
            self.bn_b = torch.nn.BatchNorm2d(64)
            self.skip = torch.nn.BatchNorm2d(8)
            self.ignore_prune_layer = ["skip"]
    net = Net().to(cuda)
    tp = pruners.dcfp_pruning(net, 0.999)
    for p in net.parameters():
        p.grad = torch.zeros_like(p)
    for step in range(4):
        for name in ("bn_a", "bn_b"):
            bn = getattr(net, name)
            bn.weight.data.copy_(torch.from_numpy(g[f"gamma:{name}:{step}"]))
            bn.weight.grad.copy_(torch.from_numpy(g[f"grad:{name}:{step}"]))
        tp.step(net)
        torch.cuda.synchronize()
        for name in ("bn_a", "bn_b"):
            mine = tp.get_eic()["eic"][name].cpu().numpy()
            assert np.array_equal(mine, g[f"eic:{name}:{step}"]), (name, step)
    assert list(tp.get_eic()["eic"].keys()) == g["names"].tolist()


def test_training_steps_vs_oracle(cuda):
    """Two full iterations (fwd, loss, bwd, EIC, SGD) with FROZEN-then-updated weights against
    the CPU oracle trainer: step 1 compares gradients/EIC/updated weights; lr is tiny so the
    trajectories stay comparable."""
    from dcfp_amd import optimizer as opt, pruners
    N, H, W = 2, 49, 65
    m = build("deeplabv3", "resnet50", True, cuda)
    sd0 = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    cfg = omodel.Cfg(model="deeplabv3", backbone="resnet50", align_corner=True)
    cpu = CpuTrainer(sd0, cfg, lr=1e-4, momentum=0.9, weight_decay=5e-4, r=0.999)

    class A:
        no_decay = None; optim = "sgd"; momentum = 0.9; learning_rate = 1e-4; weight_decay = 5e-4
    optimizer = opt.build_optimizer(A, m)
    tp = pruners.dcfp_pruning(m, 0.999)
    prev = {k: sd0[k].clone() for k in ("backbone.conv1.0.weight", "backbone.layer3.2.conv2.weight",
                                        "aspp.conv1.weight", "last_conv.6.bias", "backbone.layer2.1.bn2.weight")}
    for it in range(2):
        x = fill.closed_form_input(N, H, W) * (1.0 + 0.1 * it)
        lab = fill.closed_form_labels(N, H, W)
        optimizer.zero_grad()
        opt.adjust_learning_rate(optimizer, 1e-4, 0, 100, 0.9, -1)
        loss = m(x.to(cuda), lab.to(cuda), deepsup=True)["loss"]
        loss.backward()
        tp.step(m)
        optimizer.step()
        torch.cuda.synchronize()
        closs, _, _ = cpu.step(x, lab)
        assert abs(loss.item() - closs) < 2e-5 * max(1.0, abs(closs)), (it, loss.item(), closs)
        # EIC: compare as vectors (ReLU-mask flips move single channels: App. D) + exact-zero census
        mine = torch.cat([tp.get_eic()["eic"][n].reshape(-1) for n in cpu.scored]).cpu().numpy()
        ref = np.concatenate([cpu.eic[n].reshape(-1) for n in cpu.scored])
        rel = np.linalg.norm(mine - ref) / np.linalg.norm(ref)
        assert rel < 5e-2, (it, rel)
        assert abs(int((mine == 0).sum()) - int((ref == 0).sum())) <= 0.02 * mine.size
        # the SGD update (new - old weights) agrees within the gradient tolerance; the update
        # arithmetic itself is checked exactly in test_sgd_kernel_vs_oracle
        sd = m.state_dict()
        for k in ("backbone.conv1.0.weight", "backbone.layer3.2.conv2.weight", "aspp.conv1.weight",
                  "last_conv.6.bias", "backbone.layer2.1.bn2.weight"):
            a = sd[k].cpu().double() - prev[k].double(); b = cpu.sd[k].detach().double() - prev[k].double()
            assert ((a - b).norm() / b.norm()).item() < 5e-2, (it, k)
        prev = {k: cpu.sd[k].detach().clone() for k in prev}
        # keep both trajectories on the same weights (lr is small but errors would compound)
        m.load_state_dict({k: v.detach() for k, v in cpu.sd.items()})


def test_thirty_step_trajectory_vs_reference_golden(cuda, capsys):
    """A longer horizon than two steps (train.py:239-288 is a 4 000-step loop): 30 iterations of zero_grad -> poly LR ->
    forward -> backward -> EIC -> SGD at lr 0.01 on one fixed closed-form batch, DeepLabv3-R50 2x3x65x65, against the
    trajectory the REFERENCE's own modules produce (tests/golden/trajectory_v3_r50_2x65x65.npz, oracle/make_golden.py
    trajectory).  What only shows across steps: the persistent permuted-weight copies refreshed after every optimizer step,
    the momentum arena, pitched-buffer leases, set-to-none gradients.  (a) the loss falls as the reference's does,
    (b) step by step it stays inside the band of the reference's own fp32-vs-fp64 spread over five summation orders (tests/_parity.py), (c) the end
    state is in family with the reference's, (d) a second run reproduces the first bit for bit."""
    from dcfp_amd import optimizer as opt, pruners
    from _parity import trajectory_band
    g = np.load(os.path.join(G, "trajectory_v3_r50_2x65x65.npz"))
    N, H, W, steps = [int(v) for v in g["meta"]]
    l64, band = trajectory_band(g)

    class A:
        no_decay = None; optim = "sgd"; momentum = 0.9; learning_rate = float(g["lr0"]); weight_decay = 5e-4

    def run():
        m = build("deeplabv3", "resnet50", True, cuda)
        optimizer = opt.build_optimizer(A, m)
        tp = pruners.dcfp_pruning(m, 0.999)
        x = fill.closed_form_input(N, H, W).to(cuda)
        lab = fill.closed_form_labels(N, H, W).to(cuda)
        losses, lrs = [], []
        for it in range(steps):
            optimizer.zero_grad()
            lrs.append(opt.adjust_learning_rate(optimizer, A.learning_rate, it, int(g["max_iter"]), 0.9, -1))
            loss = m(x, lab, deepsup=True)["loss"]
            loss.backward()
            tp.step(m)
            optimizer.step()
            losses.append(loss.item())
        torch.cuda.synchronize()
        eic = torch.cat([v.reshape(-1) for v in tp.get_eic()["eic"].values()])
        return np.array(losses), np.array(lrs), eic, {k: p.detach().clone() for k, p in m.named_parameters()}, \
            list(tp.get_eic()["eic"].keys()), {k: v.clone() for k, v in m.state_dict().items()}

    losses, lrs, eic, params, eic_names, sd = run()
    assert np.allclose(lrs, g["lr64"], rtol=1e-12)
    dev = np.abs(losses - l64)
    with capsys.disabled():
        print("\n[30-step trajectory] loss %.4f -> %.4f (reference fp64 %.4f -> %.4f); largest |loss - fp64| / band: %.2f at step %d"
              % (losses[0], losses[-1], l64[0], l64[-1], float((dev / band).max()), int((dev / band).argmax())))
    # (a) the six reference runs end between 0.797 and 0.836 of their first loss
    assert 0.74 * losses[0] < losses[-1] < 0.88 * losses[0], (losses[0], losses[-1])
    # (b)
    assert (dev <= band).all(), [(t, losses[t], l64[t], band[t]) for t in np.nonzero(dev > band)[0][:5]]
    # (c) the end state: EIC vector and per-parameter weight norms as close to the fp64 reference as 3x the fp32 reference's
    # own distance (weights move by lr x 30 steps of momentum: norms agree to 1e-4; the EIC vector is 0.999-averaged)
    assert eic_names == g["eic_names"].tolist()
    e64 = g["eic64"].astype(np.float64)
    rel = np.linalg.norm(eic.double().cpu().numpy() - e64) / np.linalg.norm(e64)
    fv = [str(v) for v in g["fp32_variants"]]
    ref_rel = max(np.linalg.norm(g["eic" + v] - e64) for v in fv) / np.linalg.norm(e64)
    assert rel <= 3 * ref_rel, (rel, ref_rel)
    pn = g["param_names"].tolist()
    mine_n = np.array([float(params[k].double().norm()) for k in pn])
    n64 = g["wnorm64"]
    nerr = np.abs(mine_n - n64) / (n64 + 1e-12)
    nref = np.max([np.abs(g["wnorm" + v] - n64) for v in fv], axis=0) / (n64 + 1e-12)
    assert (nerr <= np.maximum(1e-4, 3 * nref.max())).all(), [(pn[i], nerr[i]) for i in np.argsort(-nerr)[:5]]
    rv = sd["backbone.bn1.running_var"].cpu().numpy()
    assert np.abs(rv - g["w:backbone.bn1.running_var:64"]).max() <= max(1e-5, 3 * max(np.abs(
        g["w:backbone.bn1.running_var:" + v] - g["w:backbone.bn1.running_var:64"]).max() for v in fv))
    # (d) every reduction has a fixed order and no state leaks from one step into the next run
    losses2, _, eic2, params2, _, sd2 = run()
    assert np.array_equal(losses, losses2), np.nonzero(losses != losses2)[0][:5]
    assert torch.equal(eic, eic2)
    bad = [k for k in params if not torch.equal(params[k], params2[k])]
    assert not bad, bad[:5]
    assert all(torch.equal(sd[k], sd2[k]) for k in sd)


def test_sgd_kernel_vs_oracle(cuda):
    """FusedSGD (one multi-tensor launch) vs the oracle's torch.optim.SGD restatement on
    identical gradients, two steps (first step clones the gradient into the momentum buffer)."""
    from dcfp_amd.optimizer import FusedSGD
    g = torch.Generator().manual_seed(5)
    shapes = [(64, 3, 3, 3), (19,), (70000,), (256, 128, 1, 1), (1,)]
    ps = [torch.randn(s, generator=g) for s in shapes]
    params = [torch.nn.Parameter(p.clone().to(cuda)) for p in ps]
    opt = FusedSGD([{"params": params[:3]}, {"params": params[3:], "weight_decay": 0.0}], lr=0.01,
                   momentum=0.9, weight_decay=5e-4)
    ref_p = [p.numpy().copy() for p in ps]; ref_b = [None] * len(ps)
    for step in range(2):
        grads = [torch.randn(s, generator=g) for s in shapes]
        for p, gr in zip(params, grads):
            p.grad = gr.to(cuda)
        lr = 0.01 * (1 - 0.3 * step)
        for grp in opt.param_groups:
            grp["lr"] = lr
        opt.step()
        torch.cuda.synchronize()
        for i, gr in enumerate(grads):
            wd = 5e-4 if i < 3 else 0.0
            ref_p[i], ref_b[i] = scoring.sgd_step(ref_p[i], gr.numpy(), ref_b[i], lr, 0.9, wd, first=(step == 0))
            a = params[i].detach().cpu().numpy()
            assert np.abs(a - ref_p[i]).max() <= 2e-7 * max(1.0, np.abs(ref_p[i]).max()), (step, i)


def test_pruned_model_runs_on_hip(cuda):
    """Config (5): odd channel counts after pruning go through the same HIP kernels."""
    from dcfp_amd import pruners
    from dcfp_amd.pruners.dcfp_pruner import DCFPPruner
    from oracle.make_scores import synthetic_scores
    import copy, tempfile
    m = build("deeplabv3", "resnet50", True, "cpu")
    with tempfile.TemporaryDirectory() as d:
        torch.save({"eic": synthetic_scores(m)}, d + "/score.pth")
        pr = DCFPPruner(global_percent=0.5, layer_keep=0.02, score_file=d + "/score.pth")
        pruned, cfg = pr.prune_model(copy.deepcopy(m), except_start_keys=["conv_deepsup"])
    slim = build("deeplabv3", "resnet50", True, "cpu")
    pruners.init_pruned_model(slim, cfg)
    slim.load_state_dict(pruned.state_dict())
    widths = sorted({c["out_channels"] for c in cfg.values() if "out_channels" in c})
    assert any(w % 4 for w in widths)          # genuinely odd widths
    ocfg = omodel.Cfg(model="deeplabv3", backbone="resnet50", align_corner=True)
    x, lab = fill.closed_form_input(2, 65, 65), fill.closed_form_labels(2, 65, 65)
    osd = omodel.clone_state(slim.state_dict())
    _, oloss, _ = omodel.seg_forward(osd, x, ocfg, lab, training=True)
    oloss.backward()
    slim = slim.to(cuda).train()
    loss = slim(x.to(cuda), lab.to(cuda), deepsup=True)["loss"]
    loss.backward()
    assert abs(loss.item() - float(oloss)) < 2e-5 * max(1.0, abs(float(oloss)))
    k = "backbone.layer2.0.bn2"
    a = dict(slim.named_modules())[k].weight.grad.cpu().double(); b = osd[k + ".weight"].grad.double()
    assert ((a - b).norm() / b.norm()).item() < 5e-2


def test_train_score_prune_finetune_pipeline(cuda, tmp_path):
    """Config (5) end to end with the drivers: pretrain+score (tools/train.py) -> offline prune
    loop to a FLOPs target (tools/prune.py) -> fine-tune steps on the slim model with
    --channel-cfg/--resume, all on the HIP path except the offline prune (CPU, like the reference)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import importlib
    train = importlib.import_module("train"); prune = importlib.import_module("prune")
    d1, d2, d3 = str(tmp_path / "pre"), str(tmp_path / "pr"), str(tmp_path / "ft")
    common = ["--model", "deeplabv3", "--backbone", "resnet50", "--batch-size", "2", "--input-size", "65,65",
              "--backbone-para", '{"os": 8, "mg_unit": [1, 2, 4], "inplanes": 128, "pretrained": false}']
    train.main(common + ["--num-steps", "3", "--prune-type", "dcfp", "--snapshot-dir", d1])
    assert os.path.exists(d1 + "/score.pth") and os.path.exists(d1 + "/CS_scenes_3.pth")
    score = torch.load(d1 + "/score.pth")["eic"]
    assert len(score) == 62 and all(v.dtype == torch.float32 for v in score.values())
    gp = prune.main(["--model", "deeplabv3", "--backbone", "resnet50", "--backbone-para", common[-1],
                     "--model-path", d1 + "/CS_scenes_3.pth", "--score-path", d1 + "/score.pth",
                     "--save-path", d2, "--prune-ratio", "0.4"])
    assert 0.5 <= gp < 1.0 and os.path.exists(d2 + "/channel_cfg.pth")
    train.main(common + ["--num-steps", "2", "--channel-cfg", d2 + "/channel_cfg.pth", "--resume", d2 + "/pruned.pth",
                         "--snapshot-dir", d3])
    sd = torch.load(d3 + "/CS_scenes_2.pth")
    assert all(torch.isfinite(v).all() for v in sd.values() if v.is_floating_point())


def test_inference_path_vs_oracle(cuda):
    """Eval-mode forward (BN folded into the conv epilogues), fused upsample+argmax and the
    device confusion matrix / mIoU against the CPU oracle (evaluate.py:186-247,374-380)."""
    from dcfp_amd import evaluate as ev
    from oracle import evalmetrics
    m = build("deeplabv3", "resnet50", True, cuda)
    m.criterion = None
    m.eval()
    x = fill.closed_form_input(2, 97, 129)
    lab = fill.closed_form_labels(2, 97, 129)
    cfg = omodel.Cfg(model="deeplabv3", backbone="resnet50", align_corner=True)
    osd = omodel.clone_state({k: v.cpu() for k, v in m.state_dict().items()}, requires_grad=False)
    with torch.no_grad():
        outs, _, _ = omodel.seg_forward(osd, x, cfg, None, training=False)
    logits = ev.predict_whole(m, x.to(cuda))
    # eval-mode logits with untrained running statistics are O(100): compare relative to their scale
    assert (logits.cpu() - outs[0]).abs().max().item() < 1e-4 * outs[0].abs().max().item()
    # multi-scale + flip, whole-image and sliding-window (evaluate.py:145-227), against the oracle's drivers around the
    # oracle's forward (same tiles, same flips, same resizes); tolerances relative to the logits' scale as above
    def onet(im):
        with torch.no_grad():
            return [omodel.seg_forward(osd, im, cfg, None, training=False)[0][0]]
    scale_ = outs[0].abs().max().item()
    for whole, tile in ((True, (0, 0)), (False, (65, 81))):
        ms = ev.predict_multiscale(m, x.to(cuda), tile, [0.75, 1.0], 19, True, True, whole)
        ref_ms = evalmetrics.predict_multiscale(onet, x, tile, [0.75, 1.0], 19, True, True, whole)
        assert tuple(ms.shape) == (2, 19, 97, 129)
        assert (ms.cpu() - ref_ms).abs().max().item() < 2e-4 * scale_, (whole, (ms.cpu() - ref_ms).abs().max().item(), scale_)
    # the tiling / count normalisation / flip logic alone, against the fixture produced by the reference's own functions
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "evalmetrics.npz"))
    classes, tile = int(g["classes"]), tuple(int(v) for v in g["tile"])
    img = fill.closed_form_input(2, 56, 75)[:1].to(cuda)
    pnet = evalmetrics.position_net(classes)
    assert np.abs(ev.predict_sliding(pnet, img, tile, classes).cpu().numpy() - g["sliding"]).max() < 1e-5
    assert np.abs(ev.predict_sliding(pnet, img[:, :, :20, :30], tile, classes).cpu().numpy() - g["sliding_small"]).max() < 1e-5
    for whole in (False, True):
        got = ev.predict_multiscale(pnet, img, tile, [0.75, 1.0, 1.25], classes, True, True, whole).cpu().numpy()
        assert np.abs(got - g[f"ms_whole{int(whole)}_align1"]).max() < 2e-5, whole
    pred = ev.predict_labels(m, x.to(cuda)).cpu().numpy()
    ref_pred = outs[0].argmax(1).numpy()
    top2 = outs[0].topk(2, dim=1).values
    decisive = ((top2[:, 0] - top2[:, 1]) > 1e-4 * outs[0].abs().max()).numpy()      # ignore argmax ties inside fp32 noise
    assert (pred[decisive] == ref_pred[decisive]).all() and decisive.mean() > 0.99
    cm = ev.get_confusion_matrix(lab.to(cuda), torch.from_numpy(pred).to(cuda), 19).cpu().numpy()
    keep = lab.numpy() != 255
    ref_cm = evalmetrics.confusion_matrix(lab.numpy()[keep], pred[keep], 19)
    assert np.array_equal(cm, ref_cm.astype(np.int64))
    miou, _ = ev.mean_iou(torch.from_numpy(cm))
    assert abs(miou - evalmetrics.mean_iou(ref_cm)[0]) < 1e-12


def _grads(m, x, lab):
    for p in m.parameters():
        p.grad = None
    loss = m(x, lab, deepsup=True)["loss"]
    loss.backward()
    torch.cuda.synchronize()
    return loss.item(), {k: p.grad.detach().clone() for k, p in m.named_parameters()}


def test_fused_bottleneck_matches_unfused(cuda):
    """ops.BottleneckFn (one autograd node per residual block, conv1's dgrad accumulated into the
    residual gradient, 1-bit ReLU mask) against the op-level Functions on the same weights: the same
    kernels in the same order, so every gradient must agree far below the whole-model noise floor."""
    from dcfp_amd.networks import _exec
    x = fill.closed_form_input(2, 65, 97).to(cuda); lab = fill.closed_form_labels(2, 65, 97).to(cuda)
    res = []
    for fuse in (True, False):
        _exec.FUSE_BLOCKS = fuse
        try:
            res.append(_grads(build("deeplabv3", "resnet50", True, cuda), x, lab))
        finally:
            _exec.FUSE_BLOCKS = True
    (l0, g0), (l1, g1) = res
    assert abs(l0 - l1) <= 1e-6 * abs(l1)
    worst = max(((g0[k] - g1[k]).norm() / (g1[k].norm() + 1e-30)).item() for k in g0)
    assert worst <= 1e-5, worst


def test_eval_after_training_step_refolds_bn(cuda):
    """eval -> train step -> eval in one process: the folded eval-mode BN (cached per module) must be
    rebuilt after FusedSGD / the running-statistics kernel rewrote its inputs through raw pointers."""
    from dcfp_amd import optimizer as opt
    m = build("deeplabv3", "resnet50", True, cuda)
    x = fill.closed_form_input(2, 65, 65).to(cuda); lab = fill.closed_form_labels(2, 65, 65).to(cuda)

    def eval_logits(folded):
        m.eval()
        with (torch.no_grad() if folded else torch.enable_grad()):
            return m(x, None, deepsup=True)[0].detach()
    a0 = eval_logits(True)

    class A:
        no_decay = None; optim = "sgd"; momentum = 0.9; learning_rate = 1e-2; weight_decay = 5e-4
    optimizer = opt.build_optimizer(A, m)
    m.train()
    optimizer.zero_grad()
    m(x, lab, deepsup=True)["loss"].backward()
    optimizer.step()
    a1, b1 = eval_logits(True), eval_logits(False)
    scale = b1.abs().max().item()
    assert (a1 - b1).abs().max().item() <= 1e-4 * scale
    assert (a1 - a0).abs().max().item() > 1e-2 * scale      # the step really moved the output


def _fanin_child():
    """One training step of DeepLabv3-R50 at 2x3x512x1024 (the smallest size at which the identity-shortcut blocks of
    layer1 / layer3 / layer4 reach the persistent 1x1 kernel): digests of the loss and of every parameter gradient."""
    import hashlib
    import json
    dev = torch.device("cuda:0")
    m = build("deeplabv3", "resnet50", True, dev)
    m.conv_deepsup[3].p = 0.0
    g = torch.Generator().manual_seed(21)
    x = torch.randn(2, 3, 512, 1024, generator=g).to(dev)
    lab = torch.randint(0, 19, (2, 512, 1024), generator=g).to(dev)
    loss = m(x, lab, deepsup=True)["loss"]
    loss.backward()
    torch.cuda.synchronize()
    h = hashlib.sha256()
    for _, p in m.named_parameters():
        h.update(p.grad.detach().cpu().numpy().tobytes())
    from dcfp_amd import ops
    d = ops._desc((2, 1024, 64, 128), (256, 1024, 1, 1), 1, 0, 1)
    print("FANIN_RESULT " + json.dumps({"loss": float(loss), "grads": h.hexdigest(),
                                        "layer3_ok": bool(ops.conv2d_dgrad_fanin_ok(None, torch.empty(256, 1024, 1, 1), (2, 1024, 64, 128)))}))


def test_masked_fanin_bit_identical_to_materialised_residual_gradient(cuda):
    """ops.conv2d_dgrad_fanin (conv1's dgrad adds dout * ReLU-mask from the bit mask in its epilogue) against the path
    that writes the residual gradient in the BatchNorm backward and accumulates onto it (DCFP_MASKED_FANIN=0): the same
    sums in the same order - every parameter gradient of a full step must be the same bits."""
    import json
    import subprocess
    import sys
    res = []
    for v in ("0", "1"):
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        # (DCFP_FANIN_BN_SUMS=0: the fan-in's BatchNorm-sums side output reorders bn3's sums - tested on its own below)
        env = dict(os.environ, DCFP_MASKED_FANIN=v, DCFP_FANIN_BN_SUMS="0",
                   PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--fanin-child"], env=env, capture_output=True,
                           text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        res.append(json.loads([l for l in r.stdout.splitlines() if l.startswith("FANIN_RESULT ")][-1][len("FANIN_RESULT "):]))
    assert res[1]["layer3_ok"] and not res[0]["layer3_ok"]       # the switch switches
    assert res[0]["loss"] == res[1]["loss"] and res[0]["grads"] == res[1]["grads"], res


if __name__ == "__main__" and "--fanin-child" in __import__("sys").argv:
    __import__("sys").path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    _fanin_child()


def test_fanin_epilogue_bn_sums(cuda, monkeypatch):
    """The fan-in dgrad that also reduces its result for the previous block's bn3 (ops.conv2d_dgrad_fanin_red):
    dx the same bits as the plain fan-in, the finished sums against an fp64 evaluation from the mask bits and against
    the bn_bwd_reduce kernel they replace; then a DeepLabv3-R50 step at 2x3x512x1024 with and without the coupling:
    the coupled path must actually be taken and move no parameter gradient by more than fp32 summation noise."""
    from dcfp_amd import ops
    # (without SyncBN the fused BatchNorm backward makes the epilogue sums unnecessary and they are off by default:
    #  DCFP_FANIN_BN_SUMS=2 - this switch - keeps them, which is what the data-parallel path runs)
    monkeypatch.setattr(ops, "FANIN_BN_SUMS_ALWAYS", True)
    dev = cuda
    g = torch.Generator().manual_seed(3)
    N, Cin, Cout, H, W = 2, 1024, 256, 64, 128                     # layer3 conv1 geometry at 2x3x512x1024
    xs = (N, Cin, H, W)
    rnd = lambda *sh: torch.randn(*sh, generator=g).to(dev)
    dy, w, fan_src = rnd(N, Cout, H, W), rnd(Cout, Cin, 1, 1) * 0.05, rnd(*xs)
    gam, bet = torch.rand(Cin, generator=g).to(dev) + 0.5, rnd(Cin) * 0.1
    c3, res = rnd(*xs) * 1.5 + 0.3, rnd(*xs)
    mean, var = ops.bn_stats(c3)
    _, mask = ops.bn_apply_relu_mask(c3, mean, var, gam, bet, 1e-5, res)
    c3b, resb = rnd(*xs) * 0.7 - 0.2, rnd(*xs)
    mean_b, var_b = ops.bn_stats(c3b)
    yb, mask_b = ops.bn_apply_relu_mask(c3b, mean_b, var_b, gam, bet, 1e-5, resb)
    slots = ops.conv2d_dgrad_fanin_red_slots(w, xs)
    assert slots == N * H * W // 128
    dx0 = ops.conv2d_dgrad_fanin(dy, w, xs, fan_src, mask)
    dx1, part = ops.conv2d_dgrad_fanin_red(dy, w, xs, fan_src, mask, c3b, mask_b, mean_b, slots)
    assert torch.equal(dx0, dx1)
    s1, s2, dg = ops.bn_bwd_sums_from_partials(part, var_b, 1e-5)
    gd = dx0.double() * (yb > 0).double()                          # yb = relu(...): positive exactly where the bit is set
    t1 = gd.sum(dim=(0, 2, 3))
    t2 = (gd * (c3b.double() - mean_b.double().view(1, -1, 1, 1))).sum(dim=(0, 2, 3))
    rel = lambda a, t: float((a.double() - t).norm() / t.norm())
    assert rel(s1, t1) < 2e-6 and rel(s2, t2) < 2e-6, (rel(s1, t1), rel(s2, t2))
    assert rel(dg, t2 * torch.rsqrt(var_b.double() + 1e-5)) < 2e-6
    r1, r2, _ = ops.bn_bwd_reduce(dx0, c3b, mask_b, mean_b, var_b, gam, bet, 1e-5, 3)
    assert rel(s1, r1.double()) < 2e-6 and rel(s2, r2.double()) < 2e-6

    def step(flag):
        ops.FANIN_BN_SUMS = flag
        ops.FANIN_RED_USED[0] = 0
        m = build("deeplabv3", "resnet50", True, dev)
        m.conv_deepsup[3].p = 0.0
        gg = torch.Generator().manual_seed(21)
        x = torch.randn(2, 3, 512, 1024, generator=gg).to(dev)
        lab = torch.randint(0, 19, (2, 512, 1024), generator=gg).to(dev)
        loss = m(x, lab, deepsup=True)["loss"]
        loss.backward()
        torch.cuda.synchronize()
        return float(loss), {k: p.grad.detach().clone() for k, p in m.named_parameters()}, ops.FANIN_RED_USED[0]
    saved = ops.FANIN_BN_SUMS
    try:
        l0, g0, n0 = step(False)
        l1, g1, n1 = step(True)
    finally:
        ops.FANIN_BN_SUMS = saved
    assert n0 == 0 and n1 >= 5, (n0, n1)          # layer3 of R50: 5 identity blocks behind an identity-or-downsample block
    assert l0 == l1
    a = torch.cat([v.reshape(-1).double() for v in g0.values()])
    b = torch.cat([g1[k].reshape(-1).double() for k in g0])
    assert float((a - b).norm() / a.norm()) < 1e-4          # (wrong sums would be O(1); reordered fp32 sums are ~1e-6)

    # the partial sums are only trusted for the very tensor the fan-in wrote: a gradient that reaches a block's output
    # through a hook - a new tensor with other values - must send that block's bn3 back to its own reduce kernel, and the
    # edit must arrive in the parameter gradients (same pointer + same version counter is the only state the sums are trusted in)
    from dcfp_amd.networks.backbone.resnet import Bottleneck

    def chain(hook):
        ops.FANIN_RED_USED[0] = 0
        torch.manual_seed(3)
        blocks = [Bottleneck(1024, 256, stride=1, dilation=2).to(dev).train() for _ in range(3)]
        gx = torch.Generator().manual_seed(4)
        x = torch.randn(2, 1024, 64, 128, generator=gx).to(dev).requires_grad_(True)
        h = blocks[0](x)
        if hook is not None:
            h.register_hook(hook)
        y = blocks[2](blocks[1](h))
        y.backward(torch.randn(y.shape, generator=gx).to(dev))
        torch.cuda.synchronize()
        return [p.grad.clone() for b in blocks for p in b.parameters()], ops.FANIN_RED_USED[0]
    plain, used_plain = chain(None)
    scaled, used_new = chain(lambda gr: gr * 2.0)                  # a NEW tensor arrives at block 0's output
    assert used_plain == 2 and used_new == 1, (used_plain, used_new)
    n0p = len(list(Bottleneck(1024, 256).parameters()))
    for k, (a_, b_) in enumerate(zip(plain, scaled)):
        want = a_ * 2.0 if k < n0p else a_                           # block 0 sees the doubled gradient, blocks 1 / 2 do not
        assert float((b_ - want).norm() / want.norm().clamp_min(1e-30)) < 1e-5, k


def test_dropped_graph_releases_the_pitched_buffers(cuda, monkeypatch):
    """A grad-enabled forward whose graph never runs backward (validation without no_grad, the NaN guard raising,
    train.py:260) must not leave the Bottlenecks' persistent row-pitched buffers marked busy: the next step would then
    allocate and zero-fill a fresh activation-sized buffer per block, every step, silently.  (2x3x512x1024: the size
    from which the dilation-1 / 2 convs take row-pitched operands.)"""
    from dcfp_amd import ops
    m = build("deeplabv3", "resnet50", True, cuda)
    x = fill.closed_form_input(2, 512, 1024).to(cuda)
    lab = fill.closed_form_labels(2, 512, 1024).to(cuda)
    m(x, lab, deepsup=True)["loss"].backward()               # first step creates the buffers
    owners = [mod for mod in m.modules() if getattr(mod, "_dcfp_pitch", None) is not None]
    if os.environ.get("DCFP_CONV_WINOGRAD", "1") == "0" and not owners:
        pytest.skip("direct kernels only: nothing is row-pitched at this size")
    assert len(owners) >= 5, len(owners)                     # stem BatchNorms and Bottlenecks whose 3x3 conv reads a pitched y1
    calls = []
    real = ops.new_pitched
    monkeypatch.setattr(ops, "new_pitched", lambda *a, **k: (calls.append(a[0]), real(*a, **k))[1])
    loss = m(x, lab, deepsup=True)["loss"]                   # grad-enabled forward ...
    assert calls == []
    del loss                                                 # ... whose graph is dropped without backward
    l2 = m(x, lab, deepsup=True)["loss"]
    l2.backward()
    assert calls == [], calls                                # every block found its buffer free again
    held = m(x, lab, deepsup=True)["loss"]                   # a graph that is still alive keeps its buffers:
    m(x, lab, deepsup=True)["loss"].backward()               # a second forward meanwhile gets fresh ones
    assert len(calls) > 0
    held.backward()
    torch.cuda.synchronize()
