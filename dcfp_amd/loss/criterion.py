"""Criterion factory and CE(+deep supervision) loss — API of loss/criterion.py:11-74.

`CriterionDSN.forward(preds, target)` keeps the reference contract (full-resolution logits
in, {'loss': t} out).  `forward_lowres(preds_lr, target, size, align_corner)` is the fused
entry Seg_Model.forward uses: bilinear upsample + log-softmax + NLL in one HIP kernel per
head (the 637 MB/head full-resolution logits are never written).
Both go through dcfp_upsample_ce_*; a full-resolution input is the h==H special case
(source index == destination index, lambda == 0: the interpolation is exact)."""
import torch
import torch.nn as nn

from .. import ops
from .ohem import CriterionOhemDSN  # noqa: F401  (reference imports it here, criterion.py:7)


def build_criterions(loss_type, dataset, loss_para):
    if len(loss_type.split(",")) > 1:
        return CombinedCriterion(loss_type, dataset, loss_para)
    return build_criterion(loss_type, dataset, loss_para)


def build_criterion(loss_type, dataset, loss_para):
    if loss_type == "ce":
        cls = CriterionDSN
    elif loss_type == "ohem":
        cls = CriterionOhemDSN
    elif loss_type == "gsrl":
        cls = CriterionGsrlDSN
    else:
        raise NotImplementedError(loss_type)
    return cls(dataset=dataset, **loss_para)


def class_weighted_ce(logits, target, class_weights, size, align_corner, ignore_index, pixel_keep=None):
    """nn.CrossEntropyLoss(weight=w, ignore_index, 'mean') of the upsampled logits (criterion.py:54-60):
    sum_i w[y_i] * nll_i / sum_i w[y_i] over the valid (and kept) pixels, through the fused weighted-CE
    kernels with the per-pixel weight w[y_i]."""
    w = class_weights.to(device=logits.device, dtype=torch.float32)
    valid = target != ignore_index
    if pixel_keep is not None:
        valid = valid & (pixel_keep != 0)
    pix_w = w[target.clamp(min=0, max=w.numel() - 1)] * valid
    out = ops.upsample_weighted_ce(logits, target, pix_w, size, align_corner, ignore_index)   # [N, 2] = (sum w*ce, sum w)
    return out[:, 0].sum() / out[:, 1].sum()


class CombinedCriterion(nn.Module):
    """criterion.py:30-45: sum of the 'loss' entries of several criteria."""

    def __init__(self, loss_types, dataset=None, loss_para={}):
        super().__init__()
        self.criterions = [build_criterion(t, dataset, loss_para) for t in loss_types.split(",")]

    def forward(self, preds, labels):
        loss, total = {}, 0.0
        for c in self.criterions:
            part = c(preds, labels)
            total = total + part["loss"]
            loss.update(part)
        loss["loss"] = total
        return loss

    def forward_lowres(self, preds_lr, labels, size, align_corner):
        loss, total = {}, 0.0
        for c in self.criterions:
            part = c.forward_lowres(preds_lr, labels, size, align_corner)
            total = total + part["loss"]
            loss.update(part)
        loss["loss"] = total
        return loss


class CriterionDSN(nn.Module):
    """CE(main) + ds_weight * CE(deep supervision), ignore_index = dataset.ignore_label,
    mean over valid pixels (criterion.py:48-74)."""

    def __init__(self, dataset=None, ds_weight=0.4, balance_weight=False, **kwargs):
        super().__init__()
        self.ignore_index = dataset.ignore_label
        self.ds_weight = ds_weight
        # criterion.py:54-60: nn.CrossEntropyLoss(weight=dataset.class_weights) when balance_weight
        self.class_weights = None
        if balance_weight:
            self.class_weights = torch.as_tensor(dataset.class_weights, dtype=torch.float32)

    def _ce(self, logits, target, size, align_corner):
        if self.class_weights is not None:
            return class_weighted_ce(logits, target, self.class_weights, size, align_corner, self.ignore_index)
        return ops.upsample_cross_entropy(logits, target, size, align_corner, self.ignore_index)

    def forward_lowres(self, preds, target, size, align_corner):
        if isinstance(target, dict):
            target = target["ori"]
        loss = self._ce(preds[0], target, size, align_corner)
        if len(preds) >= 2:
            loss = loss + self._ce(preds[1], target, size, align_corner) * self.ds_weight
        return {"loss": loss}

    def forward(self, preds, target):
        if isinstance(preds, dict):
            preds, target = [preds["pred"], preds["deepsup"]], target["ori"]
        size = target.shape[-2:]
        return self.forward_lowres(list(preds), target, size, True)


class CriterionGsrlDSN(nn.Module):
    """Fine-tune loss of DCFP (criterion.py:77-101): per-pixel balance weights, dilated by a k x k
    max filter and multiplied by the calibration factor 1 + gamma*(1 - (p1 - p2)) of the main
    head's top-2 softmax margin, weight CE(main) and CE(deep supervision); each is normalised
    per image by its weight sum and averaged over the batch.
    labels: {'ori': int64 [N,H,W], 'weight': float [N,H,W]}."""

    def __init__(self, dataset=None, ds_weight=0.4, k=9, gamma=9, **kwargs):
        super().__init__()
        self.k = k
        self.gamma = gamma
        self.ignore_index = dataset.ignore_label
        self.ds_weight = ds_weight

    def _term(self, logits, ori, weight, size, align_corner):
        out = ops.upsample_weighted_ce(logits, ori, weight, size, align_corner, self.ignore_index)
        return torch.mean(out[:, 0] / (out[:, 1] + 1e-8))

    def forward_lowres(self, preds, labels, size, align_corner):
        ori = labels["ori"]
        with torch.no_grad():
            weight = ops.maxfilter2d(labels["weight"].to(torch.float32), self.k)
            margin = ops.upsample_margin(preds[0].detach(), size, align_corner)
            weight = (1 + self.gamma * (1 - margin)) * weight
            weight[ori == self.ignore_index] = 0.0
        loss = self._term(preds[0], ori, weight, size, align_corner)
        if len(preds) >= 2:
            loss = loss + self.ds_weight * self._term(preds[1], ori, weight, size, align_corner)
        return {"loss": loss}

    def forward(self, preds, labels):
        return self.forward_lowres(list(preds), labels, labels["ori"].shape[-2:], True)
