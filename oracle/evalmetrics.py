"""CPU restatement of the evaluation path: confusion matrix / mIoU (evaluate.py:229-247, 374-380) and the
whole-image / sliding-window / multi-scale(+flip) prediction drivers (evaluate.py:113-117, 145-227).
Test infrastructure: pinned by tests/golden/evalmetrics.npz, which oracle/make_golden.py generates by executing the
reference's own function bodies (evaluate.py cannot be imported here: cv2)."""
from math import ceil

import numpy as np
import torch
import torch.nn.functional as F


def confusion_matrix(gt_label, pred_label, class_num):
    """evaluate.py:229-247: bincount over gt * C + pred."""
    index = (gt_label * class_num + pred_label).astype("int32")
    label_count = np.bincount(index)
    cm = np.zeros((class_num, class_num))
    for i in range(class_num):
        for j in range(class_num):
            cur = i * class_num + j
            if cur < len(label_count):
                cm[i, j] = label_count[cur]
    return cm


def mean_iou(cm):
    """evaluate.py:374-380: IoU = tp / max(1, pos + res - tp), mean over classes."""
    pos, res, tp = cm.sum(1), cm.sum(0), np.diag(cm)
    iou = tp / np.maximum(1.0, pos + res - tp)
    return iou.mean(), iou


def precision_recall(cm):
    """evaluate.py:377-378."""
    pos, res, tp = cm.sum(1), cm.sum(0), np.diag(cm)
    return (tp / (res + 1e-5)).mean(), (tp / (pos + 1e-5)).mean()


def _first(pred):
    if isinstance(pred, list):
        return pred[0]
    if isinstance(pred, dict):
        return pred["pred"]
    return pred


def pad(image, target_size):
    """evaluate.py:113-117: zero padding at the bottom / right up to the tile size."""
    rows_missing = target_size[0] - image.shape[2]
    cols_missing = target_size[1] - image.shape[3]
    return F.pad(image, (0, cols_missing, 0, rows_missing), mode="constant", value=0.0).contiguous()


def predict_sliding(net, image, tile_size, classes):
    """evaluate.py:145-184: tiles of `tile_size` at stride ceil(tile_h * 2/3) (also for the columns - the reference
    derives both from tile_size[0]), the last row / column of tiles shifted back inside the image, logits summed
    where tiles overlap and divided by the number of tiles that saw the pixel."""
    image_size = image.shape
    overlap = 1 / 3
    stride = ceil(tile_size[0] * (1 - overlap))
    tile_rows = int(ceil((image_size[2] - tile_size[0]) / stride) + 1)
    tile_cols = int(ceil((image_size[3] - tile_size[1]) / stride) + 1)
    full_probs = torch.zeros((image_size[0], classes, image_size[2], image_size[3]), dtype=image.dtype)
    count = torch.zeros((1, classes, image_size[2], image_size[3]), dtype=image.dtype)
    for row in range(tile_rows):
        for col in range(tile_cols):
            x1, y1 = int(col * stride), int(row * stride)
            x2 = min(x1 + tile_size[1], image_size[3])
            y2 = min(y1 + tile_size[0], image_size[2])
            x1 = max(int(x2 - tile_size[1]), 0)
            y1 = max(int(y2 - tile_size[0]), 0)
            img = image[:, :, y1:y2, x1:x2]
            prediction = _first(net(pad(img, tile_size)))[:, :, 0:img.shape[2], 0:img.shape[3]]
            count[0, :, y1:y2, x1:x2] += 1
            full_probs[:, :, y1:y2, x1:x2] += prediction
    full_probs /= count
    return full_probs


def predict_whole(net, image):
    """evaluate.py:186-196."""
    with torch.no_grad():
        return _first(net(image))


def predict_multiscale(net, image, tile_size, scales, classes, flip_evaluation, align_corner, whole):
    """evaluate.py:198-227: per scale: bilinear resize to int(H s) x int(W s), whole-image or sliding-window logits,
    optionally averaged with the un-flipped logits of the mirrored image, resized back and averaged over the scales."""
    N_, C_, H_, W_ = image.shape
    full_probs = torch.zeros((N_, classes, H_, W_), dtype=image.dtype)
    for scale in scales:
        scale = float(scale)
        hs, ws = int(H_ * scale), int(W_ * scale)
        scale_image = F.interpolate(image, size=[hs, ws], mode="bilinear", align_corners=align_corner)
        with torch.no_grad():
            run = (lambda im: predict_whole(net, im)) if whole else (lambda im: predict_sliding(net, im, tile_size, classes))
            scaled_probs = run(scale_image)
            if flip_evaluation:
                flip_scaled_probs = run(torch.flip(scale_image, [3]))
                scaled_probs = 0.5 * (scaled_probs + torch.flip(flip_scaled_probs, [3]))
            scaled_probs = F.interpolate(scaled_probs, size=[H_, W_], mode="bilinear", align_corners=align_corner)
        full_probs += scaled_probs
    full_probs /= len(scales)
    return full_probs


def position_net(classes):
    """A stand-in network for pinning the tiling / flipping / averaging logic without a model: pixel-wise in the image
    values but also a function of the position INSIDE the tile it is handed, so that overlapping tiles disagree and the
    count normalisation, the shifted last tiles and the zero padding all show in the result.  Works on any device."""
    def net(img):
        n, c, h, w = img.shape
        rows = torch.arange(h, dtype=img.dtype, device=img.device).view(1, 1, h, 1)
        cols = torch.arange(w, dtype=img.dtype, device=img.device).view(1, 1, 1, w)
        outs = [(k + 1) * img[:, k % c:k % c + 1] + 0.01 * (k + 1) * rows - 0.003 * cols + 0.1 * img[:, :1] * img[:, 1:2]
                for k in range(classes)]
        return [torch.cat(outs, 1)]
    return net
