"""CPU: the oracle (oracle/, a restatement of the reference's algorithm) against golden vectors
produced by importing the real reference (oracle/make_golden.py).  This is what pins the
oracle; the GPU tests then compare the HIP path with the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import fill, masks, model as omodel, ohem as oohem, scoring

G = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    path = os.path.join(G, name)
    if not os.path.exists(path):
        pytest.skip(f"{name} not generated")
    return np.load(path, allow_pickle=False)


def test_eic_trajectory_bit_exact():
    g = _load("eic_trajectory.npz")
    for name in g["names"]:
        eic = 0
        for step in range(4):
            eic = scoring.eic_step(g[f"gamma:{name}:{step}"], g[f"grad:{name}:{step}"], eic, 0.999)
            ref = g[f"eic:{name}:{step}"]
            assert eic.dtype == np.float32 and np.array_equal(eic, ref), (name, step)
        # exact zeros stay exact zeros
        assert (eic == 0).sum() == (ref == 0).sum()


def test_lr_schedule():
    g = _load("lr_schedule.npz")
    assert np.array_equal(g["poly"], np.array([scoring.lr_poly(0.01, i, 4000, 0.9) for i in (0, 1, 1999, 3999)]))
    assert np.array_equal(g["warm"], np.array([scoring.lr_warmup(0.01, i, 1000) for i in (0, 1, 500, 999, 1000)]))


def test_ohem_threshold():
    g = _load("ohem_threshold.npz")
    for tag in ("kth_le", "kth_gt", "few_valid"):
        prob = torch.softmax(torch.from_numpy(g[f"z:{tag}"]), 1).numpy()
        th = oohem.find_threshold(prob, g[f"lab:{tag}"], 255, 0.7, int(g[f"min_kept:{tag}"]))
        assert float(th) == float(g[f"th:{tag}"]), tag


MODELS = [("simple_r50_4x64x64", "simple", "resnet50"), ("v3_r50_2x65x65", "deeplabv3", "resnet50"),
          ("v3_r101_2x65x65", "deeplabv3", "resnet101")]


def product_state(model_name, backbone, align):
    """state_dict (names/shapes) of the product module tree, closed-form filled."""
    from dcfp_amd import networks
    m = getattr(networks, model_name).Seg_Model(
        backbone=backbone, backbone_para={"os": 8, "mg_unit": [1, 2, 4], "inplanes": 128, "pretrained": False},
        num_classes=19, align_corner=align, deepsup=True)
    return m, fill.closed_form_state(m.state_dict())


@pytest.mark.parametrize("tag,model_name,backbone", MODELS)
def test_state_dict_contract(tag, model_name, backbone):
    g = _load(f"model_{tag}.npz")
    m, sd = product_state(model_name, backbone, bool(g["meta"][3]))
    assert list(sd.keys()) == list(g["state_keys"])
    assert [str(tuple(v.shape)) for v in sd.values()] == list(g["state_shapes"])
    assert m.ignore_prune_layer == list(g["ignore_prune_layer"])


@pytest.mark.parametrize("tag,model_name,backbone", MODELS)
def test_oracle_model_matches_reference(tag, model_name, backbone):
    g = _load(f"model_{tag}.npz")
    N, H, W, align = [int(v) for v in g["meta"]]
    _, sd0 = product_state(model_name, backbone, bool(align))
    cfg = omodel.Cfg(model=model_name, backbone=backbone, align_corner=bool(align))
    sd = omodel.clone_state(sd0)
    x, lab = fill.closed_form_input(N, H, W), fill.closed_form_labels(N, H, W)
    outs, loss, _ = omodel.seg_forward(sd, x, cfg, lab, training=True)
    loss.backward()
    # same ATen CPU ops as the reference: agreement far below the fp32-vs-fp64 noise floor
    noise = np.abs(g["logits_d64m32"]).max()
    d = np.abs(outs[0][:, :, ::2, ::2].detach().numpy() - g["logits32"]).max()
    dds = np.abs(outs[1][:, :, ::2, ::2].detach().numpy() - g["logits_ds32"]).max()
    assert d <= max(1e-5, 0.1 * noise) and dds <= max(1e-5, 0.1 * noise), (d, dds, noise)
    assert abs(float(loss) - float(g["loss32"])) < 2e-6
    bn_w = torch.cat([sd[n + ".weight"].grad.reshape(-1) for n in g["bn_names"]]).numpy()
    ref = g["bn_wgrad32"]
    rel = np.linalg.norm(bn_w - ref) / np.linalg.norm(ref)
    noise_rel = np.linalg.norm(g["bn_wgrad64"] - ref) / np.linalg.norm(g["bn_wgrad64"])
    assert rel <= max(1e-4, 0.5 * noise_rel), (rel, noise_rel)
    for key in ("backbone.conv1.0", "backbone.layer2.0.conv2", "last_conv.6"):
        a = sd[key + ".weight"].grad.numpy(); b = g[f"wgrad:{key}:32"]
        assert np.linalg.norm(a - b) / np.linalg.norm(b) < 1e-3, key
    assert np.abs(sd["backbone.bn1.running_mean"].numpy() - g["rm:backbone.bn1:32"]).max() < 1e-6
    assert np.abs(sd["backbone.bn1.running_var"].numpy() - g["rv:backbone.bn1:32"]).max() < 1e-6


@pytest.mark.parametrize("tag,model_name,backbone,align,gp",
                         [("v3r50", "deeplabv3", "resnet50", True, 50), ("v3r50", "deeplabv3", "resnet50", True, 70),
                          ("v3r101", "deeplabv3", "resnet101", True, 50), ("simple_r50", "simple", "resnet50", False, 50)])
def test_oracle_masks_match_reference(tag, model_name, backbone, align, gp):
    """Threshold + per-layer mask arithmetic of the oracle vs DCFPPruner in the reference."""
    g = _load(f"prune_{tag}_gp{gp}.npz")
    m, sd0 = product_state(model_name, backbone, align)
    from oracle.make_scores import synthetic_scores
    eic = synthetic_scores(m)
    links = dict(zip(g["norm_conv_bn"].tolist(), g["norm_conv_conv"].tolist()))
    exc = set(g["except_layers"].tolist())
    th = masks.thresholds(eic, list(links.keys()), exc, gp / 100.0)
    assert np.array_equal(np.array([float(th[0]), float(th[1])], dtype=np.float32), g["thresh"])
    om = masks.out_masks(eic, links, exc, th, 0.02)
    groups = [set(s.split(",")) for s in g["groups"].tolist()]
    for conv, mask in om.items():
        if any(conv in grp for grp in groups):
            continue   # residual groups take the union of their members: checked in test_pruner_host
        ref = np.unpackbits(g["out:" + conv])[:mask.numel()]
        assert np.array_equal(mask.numpy().astype(np.uint8), ref), conv


def test_gsrl_oracle_matches_reference():
    import torch.nn.functional as F
    from oracle import gsrl
    g = _load("gsrl.npz")
    for tag in ("a", "b"):
        H, W, align = [int(v) for v in g[f"meta:{tag}"]]
        z0 = torch.from_numpy(g[f"z0:{tag}"]).requires_grad_(True)
        z1 = torch.from_numpy(g[f"z1:{tag}"]).requires_grad_(True)
        p = [F.interpolate(z, size=(H, W), mode="bilinear", align_corners=bool(align)) for z in (z0, z1)]
        loss = gsrl.gsrl_loss(p, torch.from_numpy(g[f"lab:{tag}"]), torch.from_numpy(g[f"wgt:{tag}"]))
        loss.backward()
        assert abs(float(loss.detach()) - float(g[f"loss:{tag}"])) < 1e-6
        assert np.abs(z0.grad.numpy() - g[f"g0:{tag}"]).max() < 1e-7


def test_eval_oracle_vs_reference_golden():
    """oracle/evalmetrics.py (sliding-window / multi-scale + flip drivers, confusion matrix, mIoU) against the outputs of
    the reference's own function bodies (evaluate.py:113-117, 145-247, 374-380; tests/golden/evalmetrics.npz)."""
    from oracle import evalmetrics as em
    g = _load("evalmetrics.npz")
    classes, tile = int(g["classes"]), tuple(int(v) for v in g["tile"])
    img = fill.closed_form_input(2, 56, 75)[:1]
    net = em.position_net(classes)
    assert np.array_equal(em.predict_sliding(net, img, tile, classes).numpy(), g["sliding"])
    assert np.array_equal(em.predict_sliding(net, img[:, :, :20, :30], tile, classes).numpy(), g["sliding_small"])
    for whole in (False, True):
        for align in (True, False):
            out = em.predict_multiscale(net, img, tile, [0.75, 1.0, 1.25], classes, True, align, whole).numpy()
            assert np.array_equal(out, g[f"ms_whole{int(whole)}_align{int(align)}"]), (whole, align)
    assert np.array_equal(em.predict_multiscale(net, img, tile, [0.5, 1.0], classes, False, True, False).numpy(), g["ms_noflip"])
    # the overlap really matters in the fixture: sliding != whole for the position-dependent stand-in network
    assert np.abs(g["ms_whole0_align1"] - g["ms_whole1_align1"]).max() > 1e-2
    cm = em.confusion_matrix(g["gt"], g["pred"], 19)
    assert np.array_equal(cm, g["cm"]) and cm.sum() == len(g["gt"]) and cm[18, 18] == 0 and cm[18, 3] > 0
    miou, iou = em.mean_iou(cm)
    assert miou == float(g["miou"]) and np.array_equal(iou, g["iou"])
    p, r = em.precision_recall(cm)
    assert p == float(g["precision"]) and r == float(g["recall"])


def test_oracle_trainer_follows_the_reference_trajectory():
    """oracle/train_step.py (the checker of the multi-step GPU tests and bench.py's CPU baseline) against the REFERENCE's
    own loop pieces (tests/golden/trajectory_v3_r50_2x65x65.npz: Seg_Model + build_optimizer + adjust_learning_rate +
    dcfp_pruning for 30 steps): first loss to 1e-6 relative, the next steps inside the band of the reference's own
    fp32 / 1-thread / fp64 spread (the summation order of this host's thread count is a fourth variant)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _parity import trajectory_band
    from oracle.train_step import CpuTrainer
    from oracle import scoring
    g = np.load(os.path.join(G, "trajectory_v3_r50_2x65x65.npz"))
    N, H, W, _ = [int(v) for v in g["meta"]]
    l64, band = trajectory_band(g)
    cfg = omodel.Cfg(model="deeplabv3", backbone="resnet50", align_corner=True)
    x, lab = fill.closed_form_input(N, H, W), fill.closed_form_labels(N, H, W)
    tr = CpuTrainer(product_state("deeplabv3", "resnet50", True)[1], cfg, lr=float(g["lr0"]), momentum=0.9, weight_decay=5e-4, r=0.999)
    for it in range(6):
        tr.lr = scoring.lr_poly(float(g["lr0"]), it, int(g["max_iter"]), 0.9)
        assert abs(tr.lr - g["lr64"][it]) <= 1e-15
        loss, _, _ = tr.step(x, lab)
        assert abs(loss - l64[it]) <= band[it], (it, loss, l64[it], band[it])
        if it == 0:
            assert abs(loss - g["loss32"][0]) <= 1e-6 * abs(loss)
