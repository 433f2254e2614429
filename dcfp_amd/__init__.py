"""dcfp_amd — MI355X (gfx950) native implementation of the DCFP segmentation-training hot path.

Python surface mirrors the reference (wzx99/DCFP): `dcfp_amd.networks`, `dcfp_amd.loss`,
`dcfp_amd.pruners`, `dcfp_amd.optimizer`, `dcfp_amd.engine`; all arithmetic runs in
libdcfp_hip.so (hand-written HIP, C-ABI in include/dcfp_hip.h).
"""
__version__ = "0.1.0"


def install_dropin():
    """Register this package's modules under the reference's top-level names
    (`networks`, `loss`, `pruners`, `optimizer`, `engine`) so train.py / prune.py style
    drivers import them unchanged."""
    import importlib
    import sys
    for name in ("networks", "loss", "pruners", "optimizer", "engine"):
        sys.modules[name] = importlib.import_module(f"dcfp_amd.{name}")
    for sub in ("networks.deeplabv3", "networks.simple", "networks.backbone",
                "networks.backbone.resnet", "networks.tools", "networks.tools.aspp",
                "loss.criterion", "loss.ohem", "pruners.dcfp_pruner", "pruners.channel_pruner"):
        try:
            sys.modules[sub] = importlib.import_module(f"dcfp_amd.{sub}")
        except ImportError:
            pass
