"""DeepLabv3 `Seg_Model` — constructor, attributes, module names and forward contract of
networks/deeplabv3.py:12-59.

forward(input, labels=None, deepsup=False):
  * with a criterion and labels -> {'loss': 0-dim tensor}.  When the criterion offers
    `forward_lowres` (CE / OHEM here) the 8x bilinear upsample is fused into the loss kernel
    and the N x C x H x W logits are never materialised (deeplabv3.py:47,50 + criterion.py:62-74);
  * otherwise -> [logits] or [logits, logits_deepsup], each N x num_classes x H x W.
"""
import torch.nn as nn

from . import _exec
from .backbone import build_backbone
from .tools.aspp import ASPP
from .. import ops

BatchNorm2d = nn.BatchNorm2d


def _head(in_ch, num_classes):
    return nn.Sequential(nn.Conv2d(in_ch, 256, kernel_size=3, stride=1, padding=1, bias=False),
                         BatchNorm2d(256), nn.ReLU(inplace=True),
                         nn.Conv2d(256, 256, kernel_size=3, stride=1, padding=1, bias=False),
                         BatchNorm2d(256), nn.ReLU(inplace=True),
                         nn.Conv2d(256, num_classes, kernel_size=1, stride=1))


def _deepsup_head(in_ch, num_classes):
    return nn.Sequential(nn.Conv2d(in_ch, 512, kernel_size=3, stride=1, padding=1, bias=False),
                         BatchNorm2d(512), nn.ReLU(inplace=True), nn.Dropout2d(0.1),
                         nn.Conv2d(512, num_classes, kernel_size=1, stride=1))


def finish(model, input, lowres, labels):
    """Shared tail of Seg_Model.forward: fused loss, or materialised full-resolution logits."""
    size = input.shape[2:]
    crit = model.criterion
    if crit is not None and labels is not None:
        if hasattr(crit, "forward_lowres"):
            return crit.forward_lowres(lowres, labels, size, model.align_corner)
        outs = [ops.upsample_bilinear(z, size, model.align_corner) for z in lowres]
        return crit(outs, labels)
    return [ops.upsample_bilinear(z, size, model.align_corner) for z in lowres]


class Seg_Model(nn.Module):
    def __init__(self, backbone="resnet", backbone_para=None, model_para=None, num_classes=21,
                 align_corner=False, criterion=None, deepsup=False, **kwards):
        super().__init__()
        backbone_para = dict(backbone_para or {})  # the reference mutates its argument (deeplabv3.py:22)
        model_para = model_para or {}
        output_stride = backbone_para.get("os", 8)
        in_channels = model_para.get("in_channels", [1024, 2048])
        self.ignore_prune_layer = model_para.get("no_prune", ["aspp.bn1"]) \
            + backbone_para.get("no_prune", ["backbone.layer4.2.bn3"])
        self.align_corner = align_corner
        backbone_para["out_index"] = [3, 4]
        self.backbone = build_backbone(backbone, backbone_para=backbone_para)
        self.aspp = ASPP(output_stride, self.align_corner, inplanes=in_channels[1])
        self.last_conv = _head(512, num_classes)
        self.criterion = criterion
        self.deepsup = deepsup
        if self.deepsup:
            self.conv_deepsup = _deepsup_head(in_channels[0], num_classes)

    def lowres_logits(self, input, deepsup=False):
        """Logits of the head(s) at 1/os resolution (before the bilinear upsample)."""
        _exec.require_device(input)
        x_deepsup, x = self.backbone(input)
        x = self.aspp(x, next_conv=self.last_conv[0])     # (the head's first 3x3 conv reads the ASPP output)
        lowres = [_exec.run_sequential(self.last_conv, x)]
        if self.deepsup and deepsup:
            lowres.append(_exec.run_sequential(self.conv_deepsup, x_deepsup))
        return lowres

    def forward(self, input, labels=None, deepsup=False):
        return finish(self, input, self.lowres_logits(input, deepsup), labels)
