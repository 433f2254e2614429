"""Evaluation helpers — the device-side half of evaluate.py:113-117,145-247,340-380: whole-image, sliding-window and
multi-scale(+flip) prediction, on-device argmax and confusion matrix, mIoU.  Inference runs the
conv kernels with the eval-mode BatchNorm folded into their epilogues (one kernel per
conv+BN+ReLU); `predict_labels` never materialises the full-resolution logits."""
from math import ceil

import torch
import torch.nn.functional as F

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


def pad(image, target_size):
    """evaluate.py:113-117: zero padding at the bottom / right up to the tile size."""
    rows_missing = target_size[0] - image.shape[2]
    cols_missing = target_size[1] - image.shape[3]
    return F.pad(image, (0, cols_missing, 0, rows_missing), mode="constant", value=0.0).contiguous()


@torch.no_grad()
def predict_sliding(net, image, tile_size, classes):
    """evaluate.py:145-184: tiles of `tile_size` with 1/3 overlap (stride ceil(tile_h * 2/3) in both directions, as the
    reference computes it), the last row / column of tiles shifted back inside the image, every tile zero-padded to the
    tile size before the network sees it.  The reference sums the tiles' logits and the per-pixel tile count on the host;
    here both accumulators stay on the device, the image is never copied back."""
    image_size = image.shape
    overlap = 1 / 3
    stride = ceil(tile_size[0] * (1 - overlap))
    tile_rows = int(ceil((image_size[2] - tile_size[0]) / stride) + 1)
    tile_cols = int(ceil((image_size[3] - tile_size[1]) / stride) + 1)
    full_probs = torch.zeros((image_size[0], classes, image_size[2], image_size[3]), device=image.device)
    count = torch.zeros((1, 1, image_size[2], image_size[3]), device=image.device)
    for row in range(tile_rows):
        for col in range(tile_cols):
            x1, y1 = int(col * stride), int(row * stride)
            x2 = min(x1 + tile_size[1], image_size[3])
            y2 = min(y1 + tile_size[0], image_size[2])
            x1 = max(int(x2 - tile_size[1]), 0)
            y1 = max(int(y2 - tile_size[0]), 0)
            img = image[:, :, y1:y2, x1:x2]
            prediction = net(pad(img, tile_size))
            if isinstance(prediction, list):
                prediction = prediction[0]
            elif isinstance(prediction, dict):
                prediction = prediction["pred"]
            count[:, :, y1:y2, x1:x2] += 1
            full_probs[:, :, y1:y2, x1:x2] += prediction[:, :, 0:img.shape[2], 0:img.shape[3]]
    full_probs /= count
    return full_probs


@torch.no_grad()
def predict_multiscale(net, image, tile_size, scales, classes, flip_evaluation, align_corner, whole=True):
    """evaluate.py:198-227 (same signature): average over the scales of the logits of the resized image - whole-image
    or sliding-window - optionally averaged with the mirrored image's, resized back to the input size."""
    N_, C_, H_, W_ = image.shape
    full = torch.zeros((N_, classes, H_, W_), device=image.device)

    def run(im):
        return predict_whole(net, im) if whole else predict_sliding(net, im, tile_size, classes)
    for scale in scales:
        scale = float(scale)
        hs, ws = int(H_ * scale), int(W_ * scale)
        img = ops.upsample_bilinear(image, (hs, ws), align_corner)
        probs = run(img)
        if flip_evaluation:
            flipped = run(torch.flip(img, [3]))
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
