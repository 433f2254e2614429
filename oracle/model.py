"""CPU restatement of the reference's Seg_Model forward (+loss) as a pure function of a
state_dict — stock ATen ops on CPU tensors, which is precisely what the reference executes
on its CPU path (it has no kernels of its own: SURVEY.md §0).

Follows: networks/backbone/resnet.py:38-58 (Bottleneck), :143-157 (ResNet.forward),
networks/tools/aspp.py:70-85 (ASPP.forward), networks/deeplabv3.py:43-59 and
networks/simple.py:47-63 (Seg_Model.forward), loss/criterion.py:62-74 (CriterionDSN).
Works in fp32 or fp64 (dtype of the tensors passed in)."""
import torch
import torch.nn.functional as F

_DEPTHS = {"50": [3, 4, 6, 3], "101": [3, 4, 23, 3], "152": [3, 8, 36, 3]}
_OS = {16: ([1, 2, 2, 1], [1, 1, 1, 2]), 8: ([1, 2, 1, 1], [1, 1, 2, 4]), 32: ([1, 2, 2, 2], [1, 1, 1, 1])}
_ASPP_D = {16: [1, 6, 12, 18], 8: [1, 12, 24, 36], 32: [1, 3, 6, 9]}


class Cfg:
    def __init__(self, model="deeplabv3", backbone="resnet50", os=8, mg_unit=(1, 2, 4),
                 align_corner=True, deepsup=True, ds_weight=0.4, ignore=255, momentum=0.1, eps=1e-5):
        self.model, self.backbone, self.os, self.mg_unit = model, backbone, os, list(mg_unit)
        self.align_corner, self.deepsup, self.ds_weight, self.ignore = align_corner, deepsup, ds_weight, ignore
        self.momentum, self.eps = momentum, eps
        for k, v in _DEPTHS.items():
            if backbone.endswith(k):
                self.layers = v


def _bn(sd, name, x, cfg, training, relu):
    """nn.BatchNorm2d train/eval (+ReLU).  Running stats in `sd` are updated in place in
    training mode like the module does (momentum 0.1, unbiased variance)."""
    rm, rv = sd[name + ".running_mean"], sd[name + ".running_var"]
    if training and (name + ".num_batches_tracked") in sd:
        sd[name + ".num_batches_tracked"] += 1
    y = F.batch_norm(x, rm, rv, sd[name + ".weight"], sd[name + ".bias"], training, cfg.momentum, cfg.eps)
    return F.relu(y) if relu else y


def _conv(sd, name, x, stride=1, pad=0, dil=1):
    return F.conv2d(x, sd[name + ".weight"], sd.get(name + ".bias"), stride, pad, dil)


def _bottleneck(sd, p, x, cfg, training, stride, dil, has_ds):
    out = _bn(sd, p + ".bn1", _conv(sd, p + ".conv1", x), cfg, training, True)
    out = _bn(sd, p + ".bn2", _conv(sd, p + ".conv2", out, stride, dil, dil), cfg, training, True)
    out = _bn(sd, p + ".bn3", _conv(sd, p + ".conv3", out), cfg, training, False)
    if has_ds:
        res = _bn(sd, p + ".downsample.1", _conv(sd, p + ".downsample.0", x, stride), cfg, training, False)
    else:
        res = x
    return F.relu(out + res)


def backbone_forward(sd, x, cfg, training):
    strides, dils = _OS[cfg.os]
    p = "backbone."
    x = _bn(sd, p + "conv1.1", _conv(sd, p + "conv1.0", x, 2, 1), cfg, training, True)
    x = _bn(sd, p + "conv1.4", _conv(sd, p + "conv1.3", x, 1, 1), cfg, training, True)
    x = _bn(sd, p + "bn1", _conv(sd, p + "conv1.6", x, 1, 1), cfg, training, True)
    x = F.max_pool2d(x, 3, 2, 1)
    feats = {}
    for li in range(1, 5):
        n = cfg.layers[li - 1] if li < 4 else len(cfg.mg_unit)
        for bi in range(n):
            d = dils[li - 1] if li < 4 else cfg.mg_unit[bi] * dils[3]
            s = strides[li - 1] if bi == 0 else 1
            has_ds = (p + f"layer{li}.{bi}.downsample.0.weight") in sd
            x = _bottleneck(sd, p + f"layer{li}.{bi}", x, cfg, training, s, d, has_ds)
        feats[li] = x
    return feats[3], feats[4]


def aspp_forward(sd, x, cfg, training):
    d = _ASPP_D[cfg.os]
    x1 = _bn(sd, "aspp.aspp1.bn", _conv(sd, "aspp.aspp1.atrous_conv", x), cfg, training, True)
    xs = [x1]
    for k in (2, 3, 4):
        xs.append(_bn(sd, f"aspp.aspp{k}.bn", _conv(sd, f"aspp.aspp{k}.atrous_conv", x, 1, d[k - 1], d[k - 1]),
                      cfg, training, True))
    g = F.adaptive_avg_pool2d(x, 1)
    g = _bn(sd, "aspp.global_avg_pool.2", _conv(sd, "aspp.global_avg_pool.1", g), cfg, training, True)
    g = F.interpolate(g, size=x.shape[2:], mode="bilinear", align_corners=cfg.align_corner)
    x = torch.cat(xs + [g], dim=1)
    return _bn(sd, "aspp.bn1", _conv(sd, "aspp.conv1", x), cfg, training, True)


def heads_forward(sd, x_ds, x, cfg, training, dropout_mask=None):
    """Low-resolution logits of both heads (before the bilinear upsample)."""
    x = _bn(sd, "last_conv.1", _conv(sd, "last_conv.0", x, 1, 1), cfg, training, True)
    x = _bn(sd, "last_conv.4", _conv(sd, "last_conv.3", x, 1, 1), cfg, training, True)
    outs = [_conv(sd, "last_conv.6", x)]
    if cfg.deepsup:
        y = _bn(sd, "conv_deepsup.1", _conv(sd, "conv_deepsup.0", x_ds, 1, 1), cfg, training, True)
        if dropout_mask is not None:   # Dropout2d(0.1) with an injected keep/scale mask
            y = y * dropout_mask.to(y.dtype).view(y.shape[0], y.shape[1], 1, 1)
        outs.append(_conv(sd, "conv_deepsup.4", y))
    return outs


def seg_forward(sd, x, cfg, labels=None, training=True, dropout_mask=None):
    """Returns (logits list at input resolution, loss or None, low-res logits list)."""
    x_ds, f = backbone_forward(sd, x, cfg, training)
    if cfg.model == "deeplabv3":
        f = aspp_forward(sd, f, cfg, training)
    lowres = heads_forward(sd, x_ds, f, cfg, training, dropout_mask)
    outs = [F.interpolate(z, size=x.shape[2:], mode="bilinear", align_corners=cfg.align_corner) for z in lowres]
    loss = None
    if labels is not None:
        loss = F.cross_entropy(outs[0], labels, ignore_index=cfg.ignore)
        if len(outs) >= 2:
            loss = loss + F.cross_entropy(outs[1], labels, ignore_index=cfg.ignore) * cfg.ds_weight
    return outs, loss, lowres


def clone_state(state_dict, dtype=None, requires_grad=True):
    """Detach-clone a state_dict to CPU (optionally another float dtype); float parameters
    (not running stats) become autograd leaves."""
    out = {}
    for k, v in state_dict.items():
        t = v.detach().cpu().clone()
        if t.is_floating_point():
            if dtype is not None:
                t = t.to(dtype)
            if requires_grad and not (k.endswith("running_mean") or k.endswith("running_var")):
                t.requires_grad_(True)
        out[k] = t
    return out
