"""Model complexity (multiply-add count, parameter count) — API of utils/flops_counter.py:35-110.

The reference (ptflops-derived) hooks every leaf module and runs a CPU forward.  The leaf
modules of dcfp_amd models are parameter holders whose forward is never called (the HIP ops
are), so the same numbers are derived statically: the known module tree is walked with shape
propagation and each supported leaf type is charged exactly what its hook would add
(flops_counter.py:400-471): conv = k*k*Cin*Cout*Hout*Wout (+ Cout*Hout*Wout with bias),
BatchNorm = 2*numel(input) (affine), ReLU = numel(output) per call, pooling = numel(input).
F.interpolate, torch.cat and Dropout2d are not counted (they are not in the reference's
module map).  Used by the prune loop to hit a target FLOPs ratio (prune.py:77-78,112-116)."""
import torch.nn as nn


def flops_to_string(flops, units="GFLOPs", precision=2):
    if units is None:
        if flops // 10**9 > 0:
            return str(round(flops / 10.**9, precision)) + " GFLOPs"
        elif flops // 10**6 > 0:
            return str(round(flops / 10.**6, precision)) + " MFLOPs"
        elif flops // 10**3 > 0:
            return str(round(flops / 10.**3, precision)) + " KFLOPs"
        return str(flops) + " FLOPs"
    scale = {"GFLOPs": 10.**9, "MFLOPs": 10.**6, "KFLOPs": 10.**3}.get(units)
    if scale:
        return str(round(flops / scale, precision)) + " " + units
    return str(flops) + " FLOPs"


def params_to_string(num_params, units=None, precision=2):
    if units is None:
        if num_params // 10**6 > 0:
            return str(round(num_params / 10**6, precision)) + " M"
        elif num_params // 10**3:
            return str(round(num_params / 10**3, precision)) + " k"
        return str(num_params)
    if units == "M":
        return str(round(num_params / 10.**6, precision)) + " " + units
    if units == "K":
        return str(round(num_params / 10.**3, precision)) + " " + units
    return str(num_params)


def get_model_parameters_number(model):
    return sum(p.numel() for p in model.parameters() if p.requires_grad)


class _Counter:
    def __init__(self):
        self.total = 0
        self.per_module = {}

    def add(self, name, flops):
        self.total += int(flops)
        self.per_module[name] = self.per_module.get(name, 0) + int(flops)

    def conv(self, name, m, shape):
        c, h, w = shape
        k, s, p, d = m.kernel_size[0], m.stride[0], m.padding[0], m.dilation[0]
        ho = (h + 2 * p - d * (k - 1) - 1) // s + 1
        wo = (w + 2 * p - d * (k - 1) - 1) // s + 1
        cout, cin = m.weight.shape[0], m.weight.shape[1]
        flops = k * k * cin * (cout // m.groups) * ho * wo
        if m.bias is not None:
            flops += cout * ho * wo
        self.add(name, flops)
        return (cout, ho, wo)

    def norm(self, name, m, shape):
        n = shape[0] * shape[1] * shape[2]
        self.add(name, 2 * n if m.affine else n)
        return shape

    def relu(self, name, shape):
        self.add(name, shape[0] * shape[1] * shape[2])
        return shape

    def pool_in(self, name, shape):
        self.add(name, shape[0] * shape[1] * shape[2])

    def sequential(self, prefix, seq, shape):
        for cname, m in seq.named_children():
            name = f"{prefix}.{cname}"
            if isinstance(m, nn.Conv2d):
                shape = self.conv(name, m, shape)
            elif isinstance(m, (nn.BatchNorm2d, nn.SyncBatchNorm)):
                shape = self.norm(name, m, shape)
            elif isinstance(m, nn.ReLU):
                shape = self.relu(name, shape)
            elif isinstance(m, nn.AdaptiveAvgPool2d):
                self.pool_in(name, shape)
                shape = (shape[0], 1, 1)
        return shape


def _count(model, input_shape, deepsup=False):
    c = _Counter()
    bb = model.backbone
    shape = c.sequential("backbone.conv1", bb.conv1, tuple(input_shape))
    shape = c.norm("backbone.bn1", bb.bn1, shape)
    shape = c.relu("backbone.relu1", shape)
    c.pool_in("backbone.maxpool", shape)
    shape = (shape[0], (shape[1] + 2 - 3) // 2 + 1, (shape[2] + 2 - 3) // 2 + 1)
    feats = {}
    for li in range(1, 5):
        for bi, blk in enumerate(getattr(bb, f"layer{li}")):
            p = f"backbone.layer{li}.{bi}"
            x_in = shape
            s = c.conv(p + ".conv1", blk.conv1, x_in); s = c.norm(p + ".bn1", blk.bn1, s); s = c.relu(p + ".relu", s)
            s = c.conv(p + ".conv2", blk.conv2, s); s = c.norm(p + ".bn2", blk.bn2, s); s = c.relu(p + ".relu", s)
            s = c.conv(p + ".conv3", blk.conv3, s); s = c.norm(p + ".bn3", blk.bn3, s)
            if blk.downsample is not None:
                c.sequential(p + ".downsample", blk.downsample, x_in)
            shape = c.relu(p + ".relu_inplace", s)
        feats[li] = shape
    x = feats[4]
    if hasattr(model, "aspp"):
        a = model.aspp
        outs = 0
        for k in (1, 2, 3, 4):
            br = getattr(a, f"aspp{k}")
            s = c.conv(f"aspp.aspp{k}.atrous_conv", br.atrous_conv, x)
            s = c.norm(f"aspp.aspp{k}.bn", br.bn, s)
            s = c.relu(f"aspp.aspp{k}.relu", s)
            outs += s[0]
        g = c.sequential("aspp.global_avg_pool", a.global_avg_pool, x)
        outs += g[0]
        x = (outs, x[1], x[2])
        if a.outplanes is not None:
            x = c.conv("aspp.conv1", a.conv1, x); x = c.norm("aspp.bn1", a.bn1, x); x = c.relu("aspp.relu", x)
    c.sequential("last_conv", model.last_conv, x)
    if deepsup and getattr(model, "deepsup", False):
        c.sequential("conv_deepsup", model.conv_deepsup, feats[3])
    return c


def get_model_complexity_info(model, input_shape, print_per_layer_stat=True, as_strings=True,
                              input_constructor=None, flush=False, ost=None):
    """Returns (flops, params) of `model` for one input of `input_shape` (C, H, W); strings
    like '123.45 GFLOPs' / '45.67 M' when as_strings (the format prune.py parses)."""
    assert type(input_shape) is tuple and len(input_shape) == 3
    assert isinstance(model, nn.Module)
    c = _count(model, input_shape, deepsup=False)   # the reference calls forward(batch): deepsup=False
    flops, params = c.total, get_model_parameters_number(model)
    if print_per_layer_stat:
        import sys
        out = ost if ost is not None else sys.stdout
        for name, f in c.per_module.items():
            print(f"{name}: {flops_to_string(f, 'GFLOPs', 3)}, {100.0 * f / max(flops, 1):.3f}% FLOPs",
                  file=out, flush=flush)
    if as_strings:
        return flops_to_string(flops), params_to_string(params)
    return flops, params
