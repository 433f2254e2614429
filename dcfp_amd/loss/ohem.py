"""OHEM cross-entropy — API of loss/ohem.py:9-119.

Reference flow: softmax on device -> full-resolution probabilities copied to the host ->
scipy zoom to 1/8 -> k-th smallest ground-truth probability via np.partition -> threshold
max(thresh, kth) -> pixels with gt-prob > threshold relabelled ignore -> CE.
Here the per-pixel ground-truth probability comes out of the fused upsample+CE forward
kernel (gt_prob), only the 1/factor-subsampled probabilities (N*H*W/64 floats) are used for
the k-th-smallest selection, and the kept-pixel mask feeds the same fused CE kernels, so the
full-resolution probability tensor never exists and nothing but one scalar is reduced."""
import torch
import torch.nn as nn

from .. import ops


def zoom_nearest_index(out_len, in_len):
    """Source index scipy.ndimage.zoom(order=0/1, mode='constant', grid_mode=False) samples
    for each output index: coordinate o * (in-1)/(out-1)  (ohem.py:22-23)."""
    if out_len <= 1:
        return torch.zeros(max(out_len, 1), dtype=torch.float64)
    return torch.arange(out_len, dtype=torch.float64) * ((in_len - 1) / (out_len - 1))


class OhemCrossEntropy2d(nn.Module):
    def __init__(self, weight=None, ignore_label=255, thresh=0.7, min_kept=100000, factor=8):
        super().__init__()
        if weight is not None:
            raise NotImplementedError("class-weighted OHEM is not on the DCFP configs")
        self.ignore_label = ignore_label
        self.thresh = float(thresh)
        self.min_kept = int(min_kept)
        self.factor = factor

    @torch.no_grad()
    def find_threshold(self, gt_prob, target):
        """Device restatement of ohem.py:20-48 on the label-class probability map
        gt_prob[N,H,W] (the reference gathers the same quantity from the zoomed softmax)."""
        N, H, W = target.shape
        f = self.factor
        h, w = int(round(H / f)), int(round(W / f))
        dev = gt_prob.device
        ys = zoom_nearest_index(h, H).to(dev)
        xs = zoom_nearest_index(w, W).to(dev)
        # labels: order-0 (nearest, round-half-even like scipy's spline order 0 == floor(x+0.5))
        yi = torch.floor(ys + 0.5).long().clamp_(0, H - 1)
        xi = torch.floor(xs + 0.5).long().clamp_(0, W - 1)
        lab = target[:, yi][:, :, xi]
        # probabilities: order-1 (bilinear) zoom of the probability map
        y0 = torch.floor(ys).long().clamp_(0, H - 1); y1 = (y0 + 1).clamp_(max=H - 1)
        x0 = torch.floor(xs).long().clamp_(0, W - 1); x1 = (x0 + 1).clamp_(max=W - 1)
        ly = (ys - y0.double()).float().view(1, -1, 1); lx = (xs - x0.double()).float().view(1, 1, -1)
        g = gt_prob
        top = g[:, y0][:, :, x0] * (1 - lx) + g[:, y0][:, :, x1] * lx
        bot = g[:, y1][:, :, x0] * (1 - lx) + g[:, y1][:, :, x1] * lx
        prob = top * (1 - ly) + bot * ly
        min_kept = self.min_kept // (f * f)
        valid = lab != self.ignore_label
        num_valid = int(valid.sum().item())
        if min_kept >= num_valid:
            return 1.0
        threshold = self.thresh
        if num_valid > 0 and min_kept > 0:
            pred = prob[valid]
            k_th = min(pred.numel(), min_kept) - 1
            kth_val = torch.kthvalue(pred, k_th + 1).values.item()
            if kth_val > self.thresh:
                threshold = kth_val
        return threshold

    def forward_lowres(self, logits, target, size, align_corner):
        out2, lse, gtp = ops.upsample_ce_forward(logits.detach(), target, size, align_corner,
                                                 self.ignore_label, want_gt_prob=True)
        threshold = self.find_threshold(gtp, target)
        keep = (gtp <= threshold)  # ohem.py:69: kept_flag = pred <= threshold
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
        if balance_weight:
            raise NotImplementedError("class-weighted OHEM is not on the DCFP configs")
        self.ignore_index = dataset.ignore_label
        self.ds_weight = ds_weight
        self.criterion1 = OhemCrossEntropy2d(None, self.ignore_index, ohem_thres, ohem_keep)

    def forward_lowres(self, preds, target, size, align_corner):
        loss = self.criterion1.forward_lowres(preds[0], target, size, align_corner)
        if len(preds) >= 2:
            loss2 = ops.upsample_cross_entropy(preds[1], target, size, align_corner, self.ignore_index)
            loss = loss + loss2 * self.ds_weight
        return {"loss": loss}

    def forward(self, preds, target):
        return self.forward_lowres(list(preds), target, target.shape[-2:], True)
