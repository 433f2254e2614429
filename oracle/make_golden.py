"""Generate tests/golden/*.npz by IMPORTING THE REAL REFERENCE (/root/reference) on CPU.

Dev-container only (the reference never travels to the GPU box).  Run:
    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/oracle/make_golden.py
Accommodations — none of them touches a reference file:
  * `ordered_set` (third-party, un-vendored) -> an insertion-ordered list stand-in;
  * torch 2.x names conv's autograd node 'ConvolutionBackward0' -> registered at run time;
  * backbone_para['pretrained'] = False (weights are downloaded files in the reference);
  * Dropout2d.p forced to 0 in the deep-supervision head for deterministic goldens.
Inputs/weights are closed-form (oracle/fill.py), so fixtures hold only outputs.
"""
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)
sys.path.insert(0, REF)
sys.dont_write_bytecode = True


class OrderedSet(list):
    def __init__(self, it=()):
        super().__init__()
        for x in it:
            self.add(x)

    def add(self, x):
        if x not in self:
            self.append(x)

    def intersection(self, o):
        return OrderedSet(x for x in self if x in o)

    def union(self, o):
        return OrderedSet(list(self) + list(o))


_m = types.ModuleType("ordered_set")
_m.OrderedSet = OrderedSet
sys.modules["ordered_set"] = _m

import networks  # noqa: E402  (reference)
import pruners  # noqa: E402
import pruners.channel_pruner as cp  # noqa: E402
import pruners.dcfp_pruner as dp  # noqa: E402
from loss.criterion import build_criterions  # noqa: E402
from loss.ohem import OhemCrossEntropy2d  # noqa: E402
import optimizer as ref_optimizer  # noqa: E402

cp.CONV += ("ConvolutionBackward",)
cp.NON_PASS = cp.CONV + cp.FC
cp.BACKWARD_PARSER_DICT["ConvolutionBackward"] = cp.ChannelPruner.conv_backward_parser

from oracle import fill  # noqa: E402
from oracle.make_scores import synthetic_scores  # noqa: E402

BB_PARA = {"os": 8, "mg_unit": [1, 2, 4], "inplanes": 128, "pretrained": False}


class _DS:
    ignore_label = 255
    num_classes = 19
    class_weights = None


def build_ref(model, backbone, align, dtype):
    crit = build_criterions("ce", _DS(), {"ds_weight": 0.4})
    cls = getattr(networks, model).Seg_Model
    m = cls(backbone=backbone, backbone_para=dict(BB_PARA), model_para={}, num_classes=19,
            align_corner=align, criterion=crit, deepsup=True)
    m.load_state_dict(fill.closed_form_state(m.state_dict()))
    m.conv_deepsup[3].p = 0.0
    return m.to(dtype)


def whole_model(tag, model, backbone, N, H, W, align):
    res = {}
    for dtype, sfx in ((torch.float32, "32"), (torch.float64, "64")):
        torch.manual_seed(0)
        m = build_ref(model, backbone, align, dtype)
        m.train()
        x = fill.closed_form_input(N, H, W, dtype)
        lab = fill.closed_form_labels(N, H, W)
        out = m(x, lab, deepsup=True)
        loss = out["loss"]
        loss.backward()
        grads = {k: p.grad.detach() for k, p in m.named_parameters()}
        bn_names = [n for n, mod in m.named_modules() if isinstance(mod, torch.nn.BatchNorm2d)]
        res["loss" + sfx] = np.array(loss.item(), dtype=np.float64)
        res["bn_wgrad" + sfx] = torch.cat([grads[n + ".weight"].reshape(-1) for n in bn_names]).numpy()
        res["bn_bgrad" + sfx] = torch.cat([grads[n + ".bias"].reshape(-1) for n in bn_names]).numpy()
        for cname in ("backbone.conv1.0", "backbone.layer1.0.conv1", "backbone.layer2.0.conv2", "last_conv.6"):
            res["wgrad:" + cname + ":" + sfx] = grads[cname + ".weight"].numpy()
        res["bgrad:last_conv.6:" + sfx] = grads["last_conv.6.bias"].numpy()
        # every parameter gradient, compactly: L2 norm and a projection on a fixed cos vector
        pnames = [k for k, _ in m.named_parameters()]
        res["grad_l2:" + sfx] = np.array([float(grads[k].double().norm()) for k in pnames])
        res["grad_proj:" + sfx] = np.array([
            float((grads[k].double().reshape(-1) * torch.cos(0.37 * torch.arange(grads[k].numel(), dtype=torch.float64))).sum())
            for k in pnames])
        if sfx == "32":
            res["param_names"] = np.array(pnames)
        # running stats after the training-mode forward
        sd = m.state_dict()
        res["rm:backbone.bn1:" + sfx] = sd["backbone.bn1.running_mean"].numpy()
        res["rv:backbone.bn1:" + sfx] = sd["backbone.bn1.running_var"].numpy()
        # logits of both heads: second (eval-free) pass in train mode would re-update stats, so
        # take them from a fresh identical model
        m2 = build_ref(model, backbone, align, dtype)
        m2.train()
        with torch.no_grad():
            outs = m2(x, None, deepsup=True)
        # stored at every 2nd pixel (fixture size); fp64 kept as the difference to fp32
        if sfx == "32":
            l32 = [o[:, :, ::2, ::2].clone() for o in outs]
            res["logits32"] = l32[0].numpy()
            res["logits_ds32"] = l32[1].numpy()
        else:
            res["logits_d64m32"] = (outs[0][:, :, ::2, ::2] - l32[0].double()).float().numpy()
            res["logits_ds_d64m32"] = (outs[1][:, :, ::2, ::2] - l32[1].double()).float().numpy()
        if sfx == "32":
            res["bn_names"] = np.array(bn_names)
            res["state_keys"] = np.array(list(sd.keys()))
            res["state_shapes"] = np.array([str(tuple(v.shape)) for v in sd.values()])
            res["ignore_prune_layer"] = np.array(m.ignore_prune_layer)
    # Round 4: the SAME fp32 reference code in four more summation orders (1 / 2 / 4 threads, oneDNN off) - only the two
    # per-tensor gradient summaries.  One fp32 run is one draw of a tensor's fp32-vs-fp64 error (which near-zero
    # pre-activations get the other ReLU mask); the per-tensor bound of tests/_parity.py takes the tensor's error as the
    # largest of the five draws.  (Appended after the arrays above: those are what they were.)
    variants = (("32t1", 1, True), ("32t2", 2, True), ("32t4", 4, True), ("32nodnn", 8, False))
    for sfx, threads, dnn in variants:
        torch.set_num_threads(threads)
        torch.backends.mkldnn.enabled = dnn
        torch.manual_seed(0)
        m = build_ref(model, backbone, align, torch.float32)
        m.train()
        x = fill.closed_form_input(N, H, W, torch.float32)
        lab = fill.closed_form_labels(N, H, W)
        m(x, lab, deepsup=True)["loss"].backward()
        grads = {k: p.grad.detach() for k, p in m.named_parameters()}
        pnames = [k for k, _ in m.named_parameters()]
        res["grad_l2:" + sfx] = np.array([float(grads[k].double().norm()) for k in pnames])
        res["grad_proj:" + sfx] = np.array([
            float((grads[k].double().reshape(-1) * torch.cos(0.37 * torch.arange(grads[k].numel(), dtype=torch.float64))).sum())
            for k in pnames])
    torch.set_num_threads(8)
    torch.backends.mkldnn.enabled = True
    res["fp32_variants"] = np.array(["32"] + [v[0] for v in variants])
    res["meta"] = np.array([N, H, W, int(align)])
    np.savez_compressed(os.path.join(OUT, f"model_{tag}.npz"), **res)
    print("wrote", tag, "loss32", res["loss32"], "loss64", res["loss64"])


def eic_trajectory():
    """dcfp_pruning.step over 4 steps on hand-made (gamma, grad) incl. sign flips, exact
    zeros and the python-int-0 start (pruners/dcfp_pruner.py:13-20)."""
    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.bn_a = torch.nn.BatchNorm2d(37)
            self.bn_b = torch.nn.BatchNorm2d(64)
            self.skip = torch.nn.BatchNorm2d(8)
            self.ignore_prune_layer = ["skip"]
    net = Net()
    tp = dp.dcfp_pruning(net, 0.999)
    rec = {}
    for step in range(4):
        for j, (name, bn) in enumerate((("bn_a", net.bn_a), ("bn_b", net.bn_b), ("skip", net.skip))):
            n = bn.weight.numel()
            i = torch.arange(n, dtype=torch.float64)
            gamma = (0.4 + torch.cos(0.9 * i + j + 0.3 * step)).float()
            grad = (1e-3 * torch.cos(1.7 * i + 2 * j + 1.1 * step) * (1 + i % 3)).float()
            grad[(i % 7 == 0)] = 0.0          # exact zeros
            if step >= 2:
                grad[(i % 5 == 1)] *= -1       # sign flips against the earlier steps
            bn.weight.data.copy_(gamma)
            bn.weight.grad = grad.clone()
            rec[f"gamma:{name}:{step}"] = gamma.numpy()
            rec[f"grad:{name}:{step}"] = grad.numpy()
        tp.step(net)
        for name in ("bn_a", "bn_b"):
            rec[f"eic:{name}:{step}"] = tp.get_eic()["eic"][name].clone().numpy()
    rec["names"] = np.array(list(tp.get_eic()["eic"].keys()))
    np.savez_compressed(os.path.join(OUT, "eic_trajectory.npz"), **rec)
    print("wrote eic_trajectory", rec["names"])


PRUNE_CASES = [("v3r50", "deeplabv3", "resnet50", True, 0.5), ("v3r50", "deeplabv3", "resnet50", True, 0.7),
               ("v3r101", "deeplabv3", "resnet101", True, 0.5), ("simple_r50", "simple", "resnet50", False, 0.5)]


def masks_and_surgery(only=None):
    for tag, model, backbone, align, gp in PRUNE_CASES:
        if only is not None and tag not in only:
            continue
        torch.manual_seed(0)
        m = build_ref(model, backbone, align, torch.float32)
        m.criterion = None
        eic = synthetic_scores(m)
        score_path = "/tmp/_golden_score.pth"
        torch.save({"eic": eic}, score_path)
        pruner = dp.DCFPPruner(global_percent=gp, layer_keep=0.02, score_file=score_path)
        import copy
        pruned, channel_cfg = pruner.prune_model(copy.deepcopy(m), except_start_keys=["conv_deepsup"])
        th = pruner.get_thresh()
        rec = {"thresh": np.array([float(th[0]), float(th[1])], dtype=np.float32),
               "names": np.array(list(channel_cfg.keys())),
               "norm_conv_bn": np.array(list(pruner.norm_conv_links.keys())),
               "norm_conv_conv": np.array(list(pruner.norm_conv_links.values())),
               "except_layers": np.array(pruner.except_layers),
               "groups": np.array([",".join(sorted(g)) for g in pruner.same_out_channel_groups.values()])}
        for name, cfg in channel_cfg.items():
            if "in_mask" in cfg:
                rec["in:" + name] = np.packbits(cfg["in_mask"].reshape(-1).astype(np.uint8))
                rec["in_n:" + name] = np.array([cfg["in_channels"], cfg["raw_in_channels"]])
            if "out_mask" in cfg:
                rec["out:" + name] = np.packbits(cfg["out_mask"].reshape(-1).astype(np.uint8))
                rec["out_n:" + name] = np.array([cfg["out_channels"], cfg["raw_out_channels"]])
        sd = pruned.state_dict()
        rec["pruned_keys"] = np.array(list(sd.keys()))
        rec["pruned_shapes"] = np.array([str(tuple(v.shape)) for v in sd.values()])
        rec["pruned_sum"] = np.array([float(v.double().sum()) for v in sd.values()])
        rec["pruned_abs"] = np.array([float(v.double().abs().sum()) for v in sd.values()])
        # slim model rebuilt by init_pruned_model + forward sanity (prune.py:100-110)
        slim = build_ref(model, backbone, align, torch.float32)
        slim.criterion = None
        pruners.init_pruned_model(slim, channel_cfg)
        rec["slim_shapes"] = np.array([str(tuple(v.shape)) for v in slim.state_dict().values()])
        slim.load_state_dict(sd)
        slim.eval()
        with torch.no_grad():
            y = slim(fill.closed_form_input(2, 33, 33), None, deepsup=True)
        rec["slim_logits"] = y[0].numpy()
        np.savez_compressed(os.path.join(OUT, f"prune_{tag}_gp{int(gp * 100)}.npz"), **rec)
        print("wrote prune golden", tag, "gp", gp, "thresh", rec["thresh"],
              "kept", sum(int(c.get("out_channels", 0)) for c in channel_cfg.values()))


def ohem_cases():
    rec = {}
    crit = OhemCrossEntropy2d(ignore_label=255, thresh=0.7, min_kept=100000)
    # (zoomed map is H/8 x W/8; min_kept is divided by 64 inside find_threshold)
    cases = {"kth_le": (2, 5, 64, 64, 0.3, 64 * 20), "kth_gt": (2, 5, 64, 64, 9.0, 64 * 3),
             "few_valid": (1, 5, 32, 32, 1.0, 100000)}
    for tag, (n, c, h, w, sharp, mk) in cases.items():
        i = torch.arange(n * c * h * w, dtype=torch.float64)
        z = (0.3 * torch.cos(2.1 * i)).reshape(n, c, h, w)
        lab = fill.closed_form_labels(n, h, w, num_classes=c)
        onehot = torch.nn.functional.one_hot(lab.clamp(max=c - 1), c).permute(0, 3, 1, 2).double()
        wob = (0.5 + 0.5 * torch.cos(0.0031 * torch.arange(n * h * w, dtype=torch.float64))).reshape(n, 1, h, w)
        if tag == "kth_gt":
            wob = 0.9 + 0.1 * wob
        z = (z + sharp * onehot * wob).float()   # label-class logit boosted by a varying margin
        prob = torch.softmax(z, 1).numpy()
        lab = lab.numpy()
        crit.min_kept = mk
        th = crit.find_threshold(prob, lab)
        rec[f"z:{tag}"] = z.numpy(); rec[f"lab:{tag}"] = lab
        rec[f"th:{tag}"] = np.array(float(th)); rec[f"min_kept:{tag}"] = np.array(crit.min_kept)
        print("ohem", tag, th)
    np.savez_compressed(os.path.join(OUT, "ohem_threshold.npz"), **rec)


def flops_golden():
    """utils/flops_counter.get_model_complexity_info on the full and the pruned v3-R50
    (prune.py:77-78,112-113), input (3,257,257) to keep the CPU forward short."""
    from utils.flops_counter import get_model_complexity_info
    import copy
    rec = {}
    for tag, bb in (("v3_r50", "resnet50"), ("v3_r101", "resnet101")):
        cls = networks.deeplabv3.Seg_Model
        m = cls(backbone=bb, backbone_para=dict(BB_PARA), model_para={}, num_classes=19, align_corner=True,
                criterion=None, deepsup=False)
        f, pcount = get_model_complexity_info(m, (3, 257, 257), print_per_layer_stat=False, as_strings=False)
        fs, ps = get_model_complexity_info(m, (3, 257, 257), print_per_layer_stat=False)
        rec[f"flops:{tag}"] = np.array(float(f)); rec[f"params:{tag}"] = np.array(float(pcount))
        rec[f"str:{tag}"] = np.array([fs, ps])
    # pruned model at global_percent 0.5
    m = build_ref("deeplabv3", "resnet50", True, torch.float32)
    m.criterion = None
    torch.save({"eic": synthetic_scores(m)}, "/tmp/_golden_score.pth")
    pruner = dp.DCFPPruner(global_percent=0.5, layer_keep=0.02, score_file="/tmp/_golden_score.pth")
    pruned, channel_cfg = pruner.prune_model(copy.deepcopy(m), except_start_keys=["conv_deepsup"])
    slim = networks.deeplabv3.Seg_Model(backbone="resnet50", backbone_para=dict(BB_PARA), model_para={},
                                        num_classes=19, align_corner=True, criterion=None, deepsup=False)
    pruners.init_pruned_model(slim, channel_cfg)
    f, pcount = get_model_complexity_info(slim, (3, 257, 257), print_per_layer_stat=False, as_strings=False)
    rec["flops:v3_r50_gp50"] = np.array(float(f)); rec["params:v3_r50_gp50"] = np.array(float(pcount))
    np.savez_compressed(os.path.join(OUT, "flops.npz"), **rec)
    print("wrote flops golden", {k: (v.tolist() if v.dtype.kind != "U" else v.tolist()) for k, v in rec.items()})


def gsrl_golden():
    """CriterionGsrlDSN (loss/criterion.py:77-101) on interpolated low-resolution logits."""
    import torch.nn.functional as F
    crit = build_criterions("gsrl", _DS(), {"ds_weight": 0.4})
    rec = {}
    for tag, (n, c, h, w, H, W, align) in {"a": (2, 19, 9, 13, 65, 97, True), "b": (2, 7, 8, 8, 33, 33, False)}.items():
        i = torch.arange(n * c * h * w, dtype=torch.float64)
        z0 = (2.0 * torch.cos(0.37 * i) + torch.cos(2.3 * i + 1)).reshape(n, c, h, w).float().requires_grad_(True)
        z1 = (1.5 * torch.cos(0.91 * i + 2)).reshape(n, c, h, w).float().requires_grad_(True)
        lab = fill.closed_form_labels(n, H, W, num_classes=c)
        j = torch.arange(n * H * W, dtype=torch.float64)
        wgt = (1.0 + 0.8 * torch.cos(0.05 * j) * (j % 7 == 0)).reshape(n, H, W).float()
        p0 = F.interpolate(z0, size=(H, W), mode="bilinear", align_corners=align)
        p1 = F.interpolate(z1, size=(H, W), mode="bilinear", align_corners=align)
        loss = crit([p0, p1], {"ori": lab, "weight": wgt})["loss"]
        loss.backward()
        rec[f"z0:{tag}"] = z0.detach().numpy(); rec[f"z1:{tag}"] = z1.detach().numpy()
        rec[f"lab:{tag}"] = lab.numpy(); rec[f"wgt:{tag}"] = wgt.numpy()
        rec[f"loss:{tag}"] = np.array(loss.item()); rec[f"g0:{tag}"] = z0.grad.numpy(); rec[f"g1:{tag}"] = z1.grad.numpy()
        rec[f"meta:{tag}"] = np.array([H, W, int(align)])
        print("gsrl", tag, loss.item())
    np.savez_compressed(os.path.join(OUT, "gsrl.npz"), **rec)


def lr_schedule():
    rec = {"poly": np.array([ref_optimizer.lr_poly(0.01, i, 4000, 0.9) for i in (0, 1, 1999, 3999)]),
           "warm": np.array([ref_optimizer.lr_warmup(0.01, i, 1000) for i in (0, 1, 500, 999, 1000)])}
    np.savez_compressed(os.path.join(OUT, "lr_schedule.npz"), **rec)


def _ref_function_source(lines, name):
    """Source text of the top-level function `name` of a reference file that cannot be imported."""
    start = next(i for i, l in enumerate(lines) if l.startswith(f"def {name}("))
    end = next((i for i in range(start + 1, len(lines)) if lines[i].startswith("def ")), len(lines))
    return "\n".join(lines[start:end])


def eval_golden():
    """evaluate.py imports cv2 / datasets and cannot be imported here; its prediction drivers and metrics are pure
    torch / numpy functions, so their own source text is executed (pad, predict_sliding, predict_whole,
    predict_multiscale, get_confusion_matrix: evaluate.py:113-117, 145-247; the mIoU lines of main(): 374-380).
    Accommodation: Tensor.cuda is the identity while they run (predict_sliding moves every tile to the GPU)."""
    from math import ceil
    import torch.nn as nn
    import torch.nn.functional as F
    from oracle import evalmetrics
    lines = open(os.path.join(REF, "evaluate.py")).read().splitlines()
    ns = {"np": np, "torch": torch, "F": F, "nn": nn, "ceil": ceil}
    for fn in ("pad", "predict_sliding", "predict_whole", "predict_multiscale", "get_confusion_matrix"):
        exec(_ref_function_source(lines, fn), ns)
    a = next(i for i, l in enumerate(lines) if l.strip().startswith("pos = confusion_matrix.sum(1)"))
    b = next(i for i in range(a, len(lines)) if lines[i].strip().startswith("mean_IU = IU_array.mean()"))
    body = "\n".join("    " + l.strip() for l in lines[a:b + 1])
    exec("def miou_lines(confusion_matrix):\n" + body + "\n    return p, r, IU_array, mean_IU\n", ns)
    classes, tile = 4, (24, 36)
    img = fill.closed_form_input(2, 56, 75)[:1]
    net = evalmetrics.position_net(classes)
    real_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a_, **k_: self
    try:
        rec = {"sliding": ns["predict_sliding"](net, img, tile, classes).numpy(),
               "sliding_small": ns["predict_sliding"](net, img[:, :, :20, :30], tile, classes).numpy()}   # image < tile
        for whole in (False, True):
            for align in (True, False):
                rec[f"ms_whole{int(whole)}_align{int(align)}"] = ns["predict_multiscale"](
                    net, img, tile, [0.75, 1.0, 1.25], classes, True, align, whole).numpy()
        rec["ms_noflip"] = ns["predict_multiscale"](net, img, tile, [0.5, 1.0], classes, False, True, False).numpy()
    finally:
        torch.Tensor.cuda = real_cuda
    g = np.random.RandomState(7)
    gt = g.randint(0, 19, size=20000)
    pred = np.where(g.rand(20000) < 0.7, gt, g.randint(0, 19, size=20000))
    pred[gt == 18] = 3                                        # a class that is never recognised (IoU 0)
    cm = ns["get_confusion_matrix"](gt, pred, 19)
    p_, r_, iu, miou = ns["miou_lines"](cm)
    rec.update(gt=gt.astype(np.int64), pred=pred.astype(np.int64), cm=cm, precision=np.float64(p_), recall=np.float64(r_),
               iou=iu, miou=np.float64(miou), tile=np.array(tile), classes=np.int64(classes))
    np.savez_compressed(os.path.join(OUT, "evalmetrics.npz"), **rec)


def trajectory_golden(steps=30):
    """The reference's training iteration (train.py:255-270: zero_grad -> adjust_learning_rate -> forward -> backward ->
    dcfp_pruning.step -> optimizer.step) run for `steps` iterations on ONE fixed closed-form batch - the reference's own
    Seg_Model, build_optimizer / adjust_learning_rate (optimizer.py:12-79) and dcfp_pruning (pruners/dcfp_pruner.py:7-26) -
    in fp32 with 8 / 4 / 2 / 1 threads and without oneDNN (summation order only) and in fp64: the loss curves and the end state.
    What a longer horizon catches and three steps do not: a stale permuted-weight copy, a momentum slip, a leaked lease."""
    class A:
        no_decay = None; optim = "sgd"; momentum = 0.9; learning_rate = 0.01; weight_decay = 5e-4
    N, H, W = 2, 65, 65
    rec = {"meta": np.array([N, H, W, steps]), "lr0": np.float64(A.learning_rate), "max_iter": np.int64(4000)}
    # fp32 variants of the SAME reference code that differ in summation order only - thread counts, and oneDNN switched off
    # (ATen's native conv: another algorithm over the same products, which is what a HIP kernel is too) - beside fp64
    variants = (("32", torch.float32, 8, True), ("32t1", torch.float32, 1, True), ("32t2", torch.float32, 2, True),
                ("32t4", torch.float32, 4, True), ("32nodnn", torch.float32, 8, False), ("64", torch.float64, 8, True))
    rec["fp32_variants"] = np.array([v[0] for v in variants if v[1] == torch.float32])
    for sfx, dtype, threads, dnn in variants:
        torch.set_num_threads(threads)
        torch.backends.mkldnn.enabled = dnn
        torch.manual_seed(0)
        m = build_ref("deeplabv3", "resnet50", True, dtype)
        m.train()
        opt_ = ref_optimizer.build_optimizer(A, m)
        tp = pruners.dcfp_pruning(m, 0.999)
        x = fill.closed_form_input(N, H, W, dtype)
        lab = fill.closed_form_labels(N, H, W)
        losses, lrs = [], []
        for it in range(steps):
            opt_.zero_grad()
            lrs.append(ref_optimizer.adjust_learning_rate(opt_, A.learning_rate, it, 4000, 0.9, -1))
            loss = m(x, lab, deepsup=True)["loss"]
            loss.backward()
            tp.step(m)
            opt_.step()
            losses.append(float(loss.item()))
        rec["loss" + sfx] = np.array(losses, dtype=np.float64)
        rec["lr" + sfx] = np.array(lrs, dtype=np.float64)
        eic = tp.get_eic()["eic"]
        rec["eic_names"] = np.array(list(eic.keys()))
        rec["eic" + sfx] = torch.cat([v.reshape(-1).float() for v in eic.values()]).numpy()
        sd = m.state_dict()
        for k in ("backbone.conv1.0.weight", "last_conv.6.bias", "backbone.layer2.1.bn2.weight", "backbone.bn1.running_var"):
            rec["w:" + k + ":" + sfx] = sd[k].float().numpy()
        # every parameter's L2 norm after the last step (a fixture stays small; norms catch a frozen or exploding tensor)
        rec["param_names"] = np.array([k for k, _ in m.named_parameters()])
        rec["wnorm" + sfx] = np.array([float(p_.detach().double().norm()) for _, p_ in m.named_parameters()])
        print("trajectory", sfx, "loss", losses[0], "->", losses[-1], "min", min(losses))
    torch.set_num_threads(8)
    torch.backends.mkldnn.enabled = True
    np.savez_compressed(os.path.join(OUT, "trajectory_v3_r50_2x65x65.npz"), **rec)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["eic", "lr", "ohem", "simple", "v3r50", "v3r101", "prune", "flops", "gsrl"]
    if "eic" in which:
        eic_trajectory()
    if "lr" in which:
        lr_schedule()
    if "ohem" in which:
        ohem_cases()
    if "simple" in which:
        whole_model("simple_r50_4x64x64", "simple", "resnet50", 4, 64, 64, False)
    if "v3r50" in which:
        whole_model("v3_r50_2x65x65", "deeplabv3", "resnet50", 2, 65, 65, True)
    if "v3r101" in which:
        whole_model("v3_r101_2x65x65", "deeplabv3", "resnet101", 2, 65, 65, True)
    if "prune" in which:
        masks_and_surgery()
    if "prune_new" in which:    # the R101 / `simple` cases added in round 2 (leaves the R50 fixtures untouched)
        masks_and_surgery(only=("v3r101", "simple_r50"))
    if "flops" in which:
        flops_golden()
    if "gsrl" in which:
        gsrl_golden()
    if "eval" in which:          # added in round 3 (leaves the other fixtures untouched)
        eval_golden()
    if "trajectory" in which:    # added in round 4 (likewise)
        trajectory_golden()
