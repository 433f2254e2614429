"""FCN-style `Seg_Model` without ASPP — networks/simple.py:11-63 (ResNet backbones only;
the HRNet branch of the reference is outside the hot path)."""
import torch.nn as nn

from . import _exec
from .backbone import build_backbone
from .deeplabv3 import _deepsup_head, _head, finish

BatchNorm2d = nn.BatchNorm2d


class Seg_Model(nn.Module):
    def __init__(self, backbone="resnet", backbone_para=None, model_para=None, num_classes=21,
                 align_corner=False, criterion=None, deepsup=False, **kwards):
        super().__init__()
        backbone_para = dict(backbone_para or {})
        model_para = model_para or {}
        in_channels = model_para.get("in_channels", [1024, 2048])
        self.ignore_prune_layer = model_para.get("no_prune", ["aspp.bn1"]) \
            + backbone_para.get("no_prune", ["backbone.layer4.2.bn3"])
        self.align_corner = align_corner
        if not backbone.startswith("resnet"):
            raise NotImplementedError(f"{backbone}: only resnet backbones are on the DCFP hot path")
        backbone_para["out_index"] = [3, 4]
        self.backbone = build_backbone(backbone, backbone_para=backbone_para)
        self.last_conv = _head(in_channels[-1], num_classes)
        self.criterion = criterion
        self.deepsup = deepsup
        if self.deepsup:
            self.conv_deepsup = _deepsup_head(in_channels[0], num_classes)

    def lowres_logits(self, input, deepsup=False):
        """Logits of the head(s) at 1/os resolution (before the bilinear upsample)."""
        _exec.require_device(input)
        x_deepsup, x = self.backbone(input)
        lowres = [_exec.run_sequential(self.last_conv, x)]
        if self.deepsup and deepsup:
            lowres.append(_exec.run_sequential(self.conv_deepsup, x_deepsup))
        return lowres

    def forward(self, input, labels=None, deepsup=False):
        return finish(self, input, self.lowres_logits(input, deepsup), labels)
