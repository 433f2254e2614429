"""ASPP head — module tree of networks/tools/aspp.py:10-85 on the HIP kernels.
Branches: 1x1, three dilated 3x3, image pooling (global mean -> 1x1 conv -> BN -> ReLU ->
broadcast), concat in the order x1..x5 (aspp.py:77), 1280 -> outplanes 1x1 + BN + ReLU.
`self.dropout` exists but is not applied, as in the reference (aspp.py:67,84)."""
import torch
import torch.nn as nn

from .. import _exec
from ... import ops

BatchNorm2d = nn.BatchNorm2d
_DILATIONS = {16: [1, 6, 12, 18], 8: [1, 12, 24, 36], 32: [1, 3, 6, 9]}


class _ASPPModule(nn.Module):
    def __init__(self, inplanes, planes, kernel_size, padding, dilation):
        super().__init__()
        self.atrous_conv = nn.Conv2d(inplanes, planes, kernel_size=kernel_size, stride=1,
                                     padding=padding, dilation=dilation, bias=False)
        self.bn = BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x):
        return _exec.conv_bn_act(self.atrous_conv, self.bn, x, relu=True)


class ASPP(nn.Module):
    def __init__(self, output_stride, align_corner, inplanes=2048, outplanes=512):
        super().__init__()
        if output_stride not in _DILATIONS:
            raise NotImplementedError
        d = _DILATIONS[output_stride]
        self.outplanes = outplanes
        self.align_corner = align_corner
        self.aspp1 = _ASPPModule(inplanes, 256, 1, padding=0, dilation=d[0])
        self.aspp2 = _ASPPModule(inplanes, 256, 3, padding=d[1], dilation=d[1])
        self.aspp3 = _ASPPModule(inplanes, 256, 3, padding=d[2], dilation=d[2])
        self.aspp4 = _ASPPModule(inplanes, 256, 3, padding=d[3], dilation=d[3])
        self.global_avg_pool = nn.Sequential(nn.AdaptiveAvgPool2d((1, 1)),
                                             nn.Conv2d(inplanes, 256, 1, stride=1, bias=False),
                                             BatchNorm2d(256), nn.ReLU(inplace=True))
        if self.outplanes is not None:
            self.conv1 = nn.Conv2d(1280, self.outplanes, 1, bias=False)
            self.bn1 = BatchNorm2d(self.outplanes)
            self.relu = nn.ReLU(inplace=True)
        self.dropout = nn.Dropout2d(0.1)

    def _fused_ok(self):
        br = [self.aspp1, self.aspp2, self.aspp3, self.aspp4]
        pool = list(self.global_avg_pool.children())
        return (all(m.atrous_conv.bias is None and m.atrous_conv.stride == (1, 1) and m.atrous_conv.groups == 1
                    for m in br) and len(pool) == 4 and isinstance(pool[1], nn.Conv2d) and pool[1].bias is None
                and pool[1].kernel_size == (1, 1))

    def forward(self, x, next_conv=None):
        if _exec.FUSE_BLOCKS and torch.is_grad_enabled() and x.requires_grad and self._fused_ok():
            # training: branches + concat as one autograd node writing channel slices (no torch.cat, input
            # gradients accumulated by the dgrad epilogue)
            br = [self.aspp1, self.aspp2, self.aspp3, self.aspp4]
            pool = list(self.global_avg_pool.children())
            tensors, bns = [], []
            for m in br:
                tensors += [m.atrous_conv.weight, m.bn.weight, m.bn.bias]
                bns.append(_exec._bn_args(m.bn))
            tensors += [pool[1].weight, pool[2].weight, pool[2].bias]
            bns.append(_exec._bn_args(pool[2]))
            cfg = {"convs": [(m.atrous_conv.padding[0], m.atrous_conv.dilation[0]) for m in br], "bn": bns}
            x = ops.aspp_branches(x, cfg, tensors)
        else:
            x1, x2, x3, x4 = self.aspp1(x), self.aspp2(x), self.aspp3(x), self.aspp4(x)
            x5 = _exec.run_sequential(self.global_avg_pool, x)
            x5 = ops.broadcast_to_hw(x5, x4.shape[2], x4.shape[3])
            x = torch.cat((x1, x2, x3, x4, x5), dim=1)
        if self.outplanes is not None:
            x = _exec.conv_bn_act(self.conv1, self.bn1, x, relu=True, next_conv=next_conv)
        return x
