"""Evaluation helpers — the device-side half of evaluate.py:186-247,340-380: whole-image and
multi-scale(+flip) prediction, on-device argmax and confusion matrix, mIoU.  Inference runs the
conv kernels with the eval-mode BatchNorm folded into their epilogues (one kernel per
conv+BN+ReLU); `predict_labels` never materialises the full-resolution logits."""
import torch

from . import ops
from .networks import _exec


@torch.no_grad()
def predict_whole(net, image):
    """evaluate.py:186-196: full-resolution logits of the main head."""
    prediction = net(image)
    if isinstance(prediction, list):
        prediction = prediction[0]
    elif isinstance(prediction, dict):
        prediction = prediction["pred"]
    return prediction


@torch.no_grad()
def predict_multiscale(net, image, scales, classes, flip_evaluation, align_corner):
    """evaluate.py:198-227 (whole-image variant): average of the logits over scales (+ flips)."""
    N_, C_, H_, W_ = image.shape
    full = torch.zeros((N_, classes, H_, W_), device=image.device)
    for scale in scales:
        scale = float(scale)
        hs, ws = int(H_ * scale), int(W_ * scale)
        img = ops.upsample_bilinear(image, (hs, ws), align_corner)
        probs = predict_whole(net, img)
        if flip_evaluation:
            flipped = predict_whole(net, torch.flip(img, [3]))
            probs = 0.5 * (probs + torch.flip(flipped, [3]))
        full += ops.upsample_bilinear(probs, (H_, W_), align_corner)
    full /= len(scales)
    return full


@torch.no_grad()
def predict_labels(net, image):
    """argmax of predict_whole without the N x C x H x W tensor: low-resolution logits of the
    main head -> fused upsample+argmax kernel -> int32 [N,H,W]."""
    _exec.require_device(image)
    lowres = net.lowres_logits(image)[0]
    return ops.upsample_argmax(lowres, image.shape[2:], net.align_corner)


def get_confusion_matrix(gt_label, pred_label, class_num, ignore_index=255, out=None):
    """evaluate.py:229-247 on the device (int64 [C,C]); pixels with gt == ignore are skipped, as the
    reference does by masking before the call (evaluate.py:344-347)."""
    return ops.confusion_matrix(pred_label.to(torch.int32), gt_label.to(torch.int64), class_num,
                                ignore_index, out)


def mean_iou(confusion_matrix):
    """evaluate.py:374-380: IoU per class = tp / (pos + res - tp); mean over classes."""
    cm = confusion_matrix.double()
    pos, res, tp = cm.sum(1), cm.sum(0), cm.diag()
    iou = tp / torch.clamp(pos + res - tp, min=1.0)
    return iou.mean().item(), iou
