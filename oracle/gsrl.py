"""CPU restatement of the GSRL fine-tune loss.  Follows loss/criterion.py:77-101."""
import torch
import torch.nn.functional as F


def gsrl_loss(preds, ori, weight, ignore=255, ds_weight=0.4, k=9, gamma=9):
    with torch.no_grad():
        w = F.max_pool2d(weight.unsqueeze(1), k, stride=1, padding=k // 2)[:, 0]
        score = torch.softmax(preds[0], 1)
        top = torch.sort(score, dim=1, descending=True)[0]
        w = (1 + gamma * (1 - (top[:, 0] - top[:, 1]))) * w
        w[ori == ignore] = 0.0
    total = 0.0
    for i, p in enumerate(preds[:2]):
        l = F.cross_entropy(p, ori, ignore_index=ignore, reduction="none")
        l = ((l * w).sum(dim=(1, 2)) / (w.sum(dim=(1, 2)) + 1e-8)).mean()
        total = total + (l if i == 0 else ds_weight * l)
    return total
