from . import resnet


def build_backbone(backbone, backbone_para=None):
    """networks/backbone/__init__.py:4-10 (HRNet is outside the hot path: SURVEY.md §2 row 16)."""
    if "resnet" in backbone:
        return resnet.build_resnet(backbone, backbone_para)
    raise NotImplementedError(backbone)
