"""Deep-stem dilated ResNet-50/101/152 — module tree and names of
networks/backbone/resnet.py:20-187, executed by the HIP kernels.

Bottleneck (resnet.py:38-58): 1x1 -> BN,ReLU -> 3x3(dil) -> BN,ReLU -> 1x1 -> BN -> (+res) -> ReLU.
The last BN, the residual add and the ReLU are ONE kernel here (bn_apply with residual).
"""
import torch.nn as nn

from .. import _exec
from ... import ops

BatchNorm2d = nn.BatchNorm2d

_DEPTHS = {"50": [3, 4, 6, 3], "101": [3, 4, 23, 3], "152": [3, 8, 36, 3]}
_OS_CFG = {16: ([1, 2, 2, 1], [1, 1, 1, 2]), 8: ([1, 2, 1, 1], [1, 1, 2, 4]),
           32: ([1, 2, 2, 2], [1, 1, 1, 1])}


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, dilation=dilation,
                               padding=dilation, bias=False)
        self.bn2 = BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, kernel_size=1, bias=False)
        self.bn3 = BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.relu_inplace = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride
        self.dilation = dilation

    def forward(self, x):
        if _exec.FUSE_BLOCKS and x.requires_grad:
            return _exec.bottleneck(self, x)   # one autograd node; residual grad fused into dgrad
        out = _exec.conv_bn_act(self.conv1, self.bn1, x, relu=True)
        out = _exec.conv_bn_act(self.conv2, self.bn2, out, relu=True)
        residual = x if self.downsample is None else _exec.run_sequential(self.downsample, x)
        return _exec.conv_bn_act(self.conv3, self.bn3, out, relu=True, residual=residual)


class ResNet(nn.Module):
    def __init__(self, block, layers, output_stride, inplanes=128, mg_unit=(1, 1, 1),
                 out_index=(1, 3, 4)):
        super().__init__()
        if output_stride not in _OS_CFG:
            raise NotImplementedError
        strides, dilations = _OS_CFG[output_stride]
        self.inplanes = inplanes
        self.out_index = list(out_index)
        self.conv1 = nn.Sequential(
            nn.Conv2d(3, 64, 3, 2, 1, bias=False), BatchNorm2d(64), nn.ReLU(inplace=True),
            nn.Conv2d(64, 64, 3, 1, 1, bias=False), BatchNorm2d(64), nn.ReLU(inplace=True),
            nn.Conv2d(64, inplanes, 3, 1, 1, bias=False))
        self.bn1 = BatchNorm2d(inplanes)
        self.relu1 = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._stage(block, 64, [dilations[0]] * layers[0], strides[0])
        self.layer2 = self._stage(block, 128, [dilations[1]] * layers[1], strides[1])
        self.layer3 = self._stage(block, 256, [dilations[2]] * layers[2], strides[2])
        # multi-grid unit: per-block dilation = mg_unit[i] * base (resnet.py:124-141)
        self.layer4 = self._stage(block, 512, [g * dilations[3] for g in mg_unit], strides[3])

    def _stage(self, block, planes, block_dilations, stride):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride,
                          bias=False),
                BatchNorm2d(planes * block.expansion))
        blocks = [block(self.inplanes, planes, stride, block_dilations[0], downsample)]
        self.inplanes = planes * block.expansion
        for d in block_dilations[1:]:
            blocks.append(block(self.inplanes, planes, dilation=d))
        return nn.Sequential(*blocks)

    def forward(self, input):
        _exec.require_device(input)
        stem = list(self.conv1.children())
        x = _exec.run_sequential(nn.Sequential(*stem[:-1]), input, tail_conv=stem[-1])   # (its input row-pitched where it wants that)
        x = _exec.conv_bn_act(stem[-1], self.bn1, x, relu=True)
        x = ops.maxpool3x3s2(x)
        outs = []
        for i in range(1, 5):
            for blk in getattr(self, "layer" + str(i)):
                x = blk(x)
            if i in self.out_index:
                if i < 4:    # this stage's output has two consumers: the head that taps it and the next stage
                    tap, x = ops.fork(x)
                    outs.append(tap)
                else:
                    outs.append(x)
        return tuple(outs)


def build_resnet(name, para):
    """resnet.py:172-187.  Pretrained weights are loaded with utils.load_model when
    para['pretrained'] is true and a checkpoint path is given via para['pretrained_path']
    (the reference's mypath.py download locations do not exist offline)."""
    para = para if para is not None else {}
    for key, layers in _DEPTHS.items():
        if name.endswith(key):
            break
    else:
        raise NotImplementedError(f"{name}: the reference builds resnet50/101/152 only")
    model = ResNet(Bottleneck, layers, para.get("os", 8), inplanes=para.get("inplanes", 128),
                   mg_unit=para.get("mg_unit", [1, 2, 4]), out_index=para.get("out_index", [1, 3, 4]))
    if para.get("pretrained", True):
        path = para.get("pretrained_path")
        if path is None:
            raise FileNotFoundError(
                "backbone_para['pretrained'] is true but no 'pretrained_path' was given; pass "
                "{'pretrained': False} for random init (reference: mypath.py:2-5 expects downloaded .pth)")
        from ...utils.pyt_utils import load_model
        load_model(model, path)
    return model
