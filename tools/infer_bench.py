#!/usr/bin/env python
"""Inference FPS of predict_whole-style evaluation (evaluate.py:314-337,365-368 prints the same
figure): DeepLabv3-R101, one 1024x2048 image per call, eval-mode BN folded into the convs,
fused upsample+argmax.  5 warm-up calls like the reference, then timed calls."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from dcfp_amd import evaluate as ev, networks  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backbone", default="resnet101")
    ap.add_argument("--size", default="1024,2048")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    h, w = [int(v) for v in a.size.split(",")]
    bb = {"os": 8, "mg_unit": [1, 2, 4], "inplanes": 128, "pretrained": False}
    m = networks.deeplabv3.Seg_Model(backbone=a.backbone, backbone_para=bb, num_classes=19, align_corner=True,
                                     deepsup=False).to(dev).eval()
    x = torch.randn(a.batch, 3, h, w, device=dev)
    for _ in range(5):
        ev.predict_labels(m, x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        ev.predict_labels(m, x)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"inference: {a.batch * a.iters / dt:.2f} images/s ({dt / a.iters * 1e3:.1f} ms per batch of {a.batch}) "
          f"DeepLabv3-{a.backbone} {h}x{w} fp32" + (" (conv math: bf16x3 split)" if os.environ.get("DCFP_CONV_MATH") == "bf16x3" else ""))


if __name__ == "__main__":
    main()
