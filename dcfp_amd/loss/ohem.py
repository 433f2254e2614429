"""OHEM cross-entropy — API of loss/ohem.py:9-119.

Reference flow: softmax on device -> FULL-resolution probabilities copied to the host ->
scipy zoom to 1/8 (order 1 for probabilities, order 0 for labels) -> k-th smallest
ground-truth probability via np.partition -> threshold = max(thresh, kth) -> pixels whose
ground-truth probability exceeds the threshold are relabelled ignore -> CE.
Here nothing full-resolution but one float per pixel exists: the fused upsample+CE forward
kernel yields the per-pixel LSE and label-class probability, `dcfp_ohem_zoom_gt_prob_f32`
evaluates the zoomed ground-truth probability on the 1/8 grid with scipy's coordinate rule,
the k-th smallest of those ~N*H*W/64 values is selected on the device, and the kept-pixel mask
feeds the same fused CE kernels."""
import ctypes as C

import torch
import torch.nn as nn

from .. import _lib, ops
from .._lib import check


class OhemCrossEntropy2d(nn.Module):
    def __init__(self, weight=None, ignore_label=255, thresh=0.7, min_kept=100000, factor=8):
        super().__init__()
        # class weights of the final CrossEntropyLoss (ohem.py:18); the threshold search does not use them
        self.class_weights = None if weight is None else torch.as_tensor(weight, dtype=torch.float32)
        self.ignore_label = ignore_label
        self.thresh = float(thresh)
        self.min_kept = int(min_kept)
        self.factor = factor

    @torch.no_grad()
    def find_threshold(self, logits, target, lse, size, align_corner):
        """ohem.py:20-48; returns a python float like the reference (one host sync: not used by the
        training path, which keeps the threshold on the device - threshold_device)."""
        return float(self.threshold_device(logits, target, lse, size, align_corner).item())

    @torch.no_grad()
    def threshold_device(self, logits, target, lse, size, align_corner):
        """ohem.py:20-48 entirely on the device: zoomed ground-truth probabilities -> exact radix select
        of the k-th smallest valid one -> max(thresh, kth) as a one-float device tensor."""
        logits = logits.contiguous()
        N, Cc, h, w = logits.shape
        H, W = int(size[0]), int(size[1])
        f = self.factor
        H8, W8 = int(round(H * (1.0 / f))), int(round(W * (1.0 / f)))   # scipy: round(in * zoom)
        pred8 = torch.empty((N, H8, W8), dtype=torch.float32, device=logits.device)
        lab8 = torch.empty((N, H8, W8), dtype=torch.int32, device=logits.device)
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(_lib.lib().dcfp_ohem_zoom_gt_prob_f32(
            C.c_void_p(logits.data_ptr()), C.c_void_p(target.data_ptr()), C.c_void_p(lse.data_ptr()),
            N, Cc, h, w, H, W, int(bool(align_corner)), H8, W8, C.c_void_p(pred8.data_ptr()),
            C.c_void_p(lab8.data_ptr()), stream), "ohem_zoom")
        thr = torch.empty(1, dtype=torch.float32, device=logits.device)
        check(_lib.lib().dcfp_ohem_threshold_f32(
            C.c_void_p(pred8.data_ptr()), C.c_void_p(lab8.data_ptr()), pred8.numel(), int(self.ignore_label),
            float(self.thresh), int(self.min_kept // (f * f)), C.c_void_p(thr.data_ptr()), stream),
            "ohem_threshold")
        return thr

    def forward_lowres(self, logits, target, size, align_corner):
        target = target.contiguous()
        out2, lse, gtp = ops.upsample_ce_forward(logits.detach(), target, size, align_corner,
                                                 self.ignore_label, want_gt_prob=True)
        thr = self.threshold_device(logits.detach(), target, lse, size, align_corner)
        keep = torch.empty(gtp.shape, dtype=torch.uint8, device=gtp.device)
        check(_lib.lib().dcfp_ohem_keep_mask_u8(   # ohem.py:69: kept_flag = pred <= threshold
            C.c_void_p(gtp.data_ptr()), C.c_void_p(thr.data_ptr()), gtp.numel(), C.c_void_p(keep.data_ptr()),
            C.c_void_p(torch.cuda.current_stream().cuda_stream)), "ohem_keep_mask")
        if self.class_weights is not None:
            from .criterion import class_weighted_ce
            return class_weighted_ce(logits, target, self.class_weights, size, align_corner, self.ignore_label, keep)
        return ops.upsample_cross_entropy(logits, target, size, align_corner, self.ignore_label,
                                          pixel_keep=keep)

    def forward(self, predict, target, weight=None):
        assert not target.requires_grad
        return self.forward_lowres(predict, target, target.shape[-2:], True)


class CriterionOhemDSN(nn.Module):
    """OHEM-CE(main) + ds_weight * CE(deep supervision) (ohem.py:95-119)."""

    def __init__(self, dataset=None, ds_weight=0.4, balance_weight=False, ohem_thres=0.7,
                 ohem_keep=100000, **kwargs):
        super().__init__()
        self.ignore_index = dataset.ignore_label
        self.ds_weight = ds_weight
        weight = dataset.class_weights if balance_weight else None        # ohem.py:105-108
        self.class_weights = None if weight is None else torch.as_tensor(weight, dtype=torch.float32)
        self.criterion1 = OhemCrossEntropy2d(weight, self.ignore_index, ohem_thres, ohem_keep)

    def forward_lowres(self, preds, target, size, align_corner):
        loss = self.criterion1.forward_lowres(preds[0], target, size, align_corner)
        if len(preds) >= 2:
            if self.class_weights is not None:
                from .criterion import class_weighted_ce
                loss2 = class_weighted_ce(preds[1], target, self.class_weights, size, align_corner, self.ignore_index)
            else:
                loss2 = ops.upsample_cross_entropy(preds[1], target, size, align_corner, self.ignore_index)
            loss = loss + loss2 * self.ds_weight
        return {"loss": loss}

    def forward(self, preds, target):
        return self.forward_lowres(list(preds), target, target.shape[-2:], True)
