"""Execution helpers: run the reference's leaf modules (nn.Conv2d / nn.BatchNorm2d / nn.ReLU /
nn.Dropout2d / nn.AdaptiveAvgPool2d — kept as plain torch modules so names, state_dict keys
and `isinstance` checks in pruners/flops counters stay valid) through the HIP kernels.
The leaf modules are parameter holders only: their own forward() is never called here."""
import torch
import torch.nn as nn

from .. import ops

_BN_TYPES = (nn.BatchNorm2d, nn.SyncBatchNorm)


def conv(m, x):
    if m.groups != 1 or m.kernel_size[0] != m.kernel_size[1] or m.stride[0] != m.stride[1] \
            or m.padding[0] != m.padding[1] or m.dilation[0] != m.dilation[1] \
            or isinstance(m.padding, str) or m.padding_mode != "zeros":
        raise RuntimeError(f"dcfp_amd: unsupported conv configuration {m}")
    return ops.conv2d(x, m.weight, m.bias, m.stride[0], m.padding[0], m.dilation[0])


import os

FUSE_BLOCKS = os.environ.get("DCFP_NO_BLOCK_FUSION") is None


def _bn_args(m):
    """(running_mean, running_var, training, momentum, eps, sync, num_batches_tracked) of a BN module.
    The module-side bookkeeping of nn.BatchNorm2d.forward (num_batches_tracked += 1) is done by the HIP
    kernel that finalises the batch statistics, which is handed the buffer; only the cumulative-average
    mode (momentum=None: the factor depends on the counter's value) does it here, with a host read."""
    training = m.training or (m.running_mean is None)
    sync = False
    if isinstance(m, nn.SyncBatchNorm) and training:
        sync = m.process_group if m.process_group is not None else True
    momentum = m.momentum
    nbt = None
    if training:
        m._dcfp_fold = None     # running statistics are about to change through raw pointers
    if training and m.track_running_stats and m.num_batches_tracked is not None:
        if momentum is None:
            m.num_batches_tracked.add_(1)
            momentum = 1.0 / float(m.num_batches_tracked)
        else:
            nbt = m.num_batches_tracked
    if not m.affine:
        raise RuntimeError("dcfp_amd: BatchNorm without affine parameters is not on the DCFP path")
    return (m.running_mean, m.running_var, training, momentum, m.eps, sync, nbt)


def _conv_pitch(cm, xshape):
    """Row pitch for the shifted operands of conv module `cm` on an input of `xshape` (0: dense)."""
    if cm is None or cm.kernel_size != (3, 3) or cm.groups != 1 or isinstance(cm.padding, str):
        return 0
    return ops.conv_pitch(tuple(xshape), tuple(cm.weight.shape), cm.stride[0], cm.padding[0], cm.dilation[0])


def bn_act(m, x, relu=False, residual=None, prev_conv=None, next_conv=None):
    """BatchNorm2d / SyncBatchNorm (+ReLU) (+residual add before the ReLU).  prev_conv / next_conv: the Conv2d
    modules that produced x / will read y; where they are dilation-1/2 3x3 convs on the 256 x 256-tile kernels,
    dx / y are written row-pitched for them (ops.conv_pitch)."""
    rm, rv, training, momentum, eps, sync, nbt = _bn_args(m)
    pitch_cfg = None
    if x.is_cuda and torch.is_grad_enabled() and x.requires_grad and x.dim() == 4:
        y_pitch = _conv_pitch(next_conv, x.shape) if residual is None else 0
        dx_pitch = 0
        if prev_conv is not None and prev_conv.kernel_size == (3, 3):
            in_shape = (x.shape[0], prev_conv.in_channels, x.shape[2], x.shape[3])    # stride 1, same size if eligible
            dx_pitch = _conv_pitch(prev_conv, in_shape)
        if y_pitch or dx_pitch:
            pitch_cfg = (m, y_pitch, dx_pitch)
    return ops.batch_norm_act(x, m.weight, m.bias, rm, rv, residual, relu, training, momentum, eps, sync, nbt, pitch_cfg)


def _fold(bn):
    """(scale, shift) of an eval-mode BatchNorm, cached until its tensors change.  The HIP optimizer
    and the running-statistics kernel write through raw pointers (no `_version` bump), so the key
    also carries ops.WEIGHT_EPOCH (bumped by every FusedSGD.step / load_state_dict) and a training
    forward drops the cache (_bn_args)."""
    key = (bn.weight._version, bn.bias._version, bn.running_mean._version, bn.running_var._version,
           bn.weight.data_ptr(), bn.running_mean.data_ptr(), ops.WEIGHT_EPOCH[0])
    cached = getattr(bn, "_dcfp_fold", None)
    if cached is None or cached[0] != key:
        with torch.no_grad():
            scale = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
            shift = bn.bias - bn.running_mean * scale
        cached = (key, scale.contiguous(), shift.contiguous())
        bn._dcfp_fold = cached
    return cached[1], cached[2]


def conv_bn_act(cm, bn, x, relu=False, residual=None, next_conv=None):
    """conv -> BatchNorm (+residual) (+ReLU).  Inference (BN in eval mode, no autograd): ONE kernel,
    the BN folded into the conv epilogue; otherwise conv + the training BN kernels."""
    if (not bn.training) and bn.running_mean is not None and not torch.is_grad_enabled() \
            and cm.bias is None and cm.groups == 1 and cm.padding_mode == "zeros":
        scale, shift = _fold(bn)
        return ops.conv2d_fused_infer(x, cm.weight, scale, shift, cm.stride[0], cm.padding[0],
                                      cm.dilation[0], residual, relu)
    return bn_act(bn, conv(cm, x), relu=relu, residual=residual, prev_conv=cm, next_conv=next_conv)


def _plain_conv(m, k):
    ok = (m.groups == 1 and m.bias is None and m.kernel_size == (k, k) and m.padding_mode == "zeros"
          and not isinstance(m.padding, str))
    if not ok:
        raise RuntimeError(f"dcfp_amd: unsupported conv configuration in Bottleneck: {m}")


def bottleneck(blk, x):
    """Run a reference Bottleneck (resnet.py:38-58) as the fused ops.BottleneckFn node."""
    _plain_conv(blk.conv1, 1); _plain_conv(blk.conv2, 3); _plain_conv(blk.conv3, 1)
    bns = [blk.bn1, blk.bn2, blk.bn3]
    tensors = [blk.conv1.weight, blk.bn1.weight, blk.bn1.bias, blk.conv2.weight, blk.bn2.weight,
               blk.bn2.bias, blk.conv3.weight, blk.bn3.weight, blk.bn3.bias]
    stride, dil = blk.conv2.stride[0], blk.conv2.dilation[0]
    if blk.conv2.padding[0] != dil or blk.conv1.stride[0] != 1 or blk.conv3.stride[0] != 1:
        raise RuntimeError("dcfp_amd: unexpected Bottleneck geometry")
    if blk.downsample is not None:
        dconv, dbn = blk.downsample[0], blk.downsample[1]
        _plain_conv(dconv, 1)
        if dconv.stride[0] != stride:
            raise RuntimeError("dcfp_amd: downsample stride differs from conv2 stride")
        tensors += [dconv.weight, dbn.weight, dbn.bias]
        bns.append(dbn)
    cfg = {"stride": stride, "dil": dil, "bn": [_bn_args(b) for b in bns], "owner": blk}
    return ops.bottleneck(x, cfg, tensors)


def run_sequential(seq, x, tail_conv=None):
    """Interpret an nn.Sequential of the reference (stem, downsample, heads, image-pool branch)
    with conv -> (BN [+ReLU]) peephole fusion.  tail_conv: the conv module the caller runs on the result."""
    mods = list(seq.children())
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, nn.Conv2d) and i + 1 < len(mods) and isinstance(mods[i + 1], _BN_TYPES):
            fuse = i + 2 < len(mods) and isinstance(mods[i + 2], nn.ReLU)
            nxt = None                                                    # conv -> BN -> ReLU -> conv: y feeds it directly
            if fuse:
                nxt = mods[i + 3] if i + 3 < len(mods) else (tail_conv if i + 3 == len(mods) else None)
            x = conv_bn_act(m, mods[i + 1], x, relu=fuse, next_conv=nxt if isinstance(nxt, nn.Conv2d) else None)
            i += 2 if fuse else 1
        elif isinstance(m, nn.Conv2d):
            x = conv(m, x)
        elif isinstance(m, _BN_TYPES):
            fuse = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
            x = bn_act(m, x, relu=fuse)
            if fuse:
                i += 1
        elif isinstance(m, nn.ReLU):
            raise RuntimeError("dcfp_amd: bare ReLU outside a BN+ReLU pair is not on the DCFP path")
        elif isinstance(m, nn.Dropout2d):
            x = ops.dropout2d(x, m.p, m.training, getattr(m, "fixed_mask", None))
        elif isinstance(m, nn.AdaptiveAvgPool2d):
            x = ops.global_avg_pool(x)
        elif isinstance(m, nn.Identity):
            pass
        else:
            raise RuntimeError(f"dcfp_amd: no HIP path for module {type(m).__name__}")
        i += 1
    return x


def require_device(x):
    if not x.is_cuda:
        raise RuntimeError(
            "dcfp_amd networks run on the MI355X HIP kernels only: move the model and inputs to "
            "cuda (the CPU restatement used for checking lives under oracle/, not in the product)")
