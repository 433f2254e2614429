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
    else:
        # 'gsrl' (criterion.py:77-101) belongs to the fine-tune stage: SURVEY.md §8(f) rank 3
        raise NotImplementedError(loss_type)
    return cls(dataset=dataset, **loss_para)


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
        if balance_weight:
            raise NotImplementedError("class-weighted CE (balance_weight) is not on the DCFP configs")

    def _ce(self, logits, target, size, align_corner):
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
