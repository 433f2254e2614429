"""One CPU training iteration of the reference's loop (train.py:255-270) over the oracle
model: zero_grad -> forward -> CE + 0.4*CE -> backward -> EIC step -> SGD step.
Used as the checker for whole-step parity and as bench.py's timed CPU baseline ("port")."""
import numpy as np
import torch

from . import model as omodel
from . import scoring


class CpuTrainer:
    def __init__(self, state_dict, cfg, ignore_prune=("aspp.bn1", "backbone.layer4.2.bn3"),
                 lr=0.01, momentum=0.9, weight_decay=5e-4, r=0.999, dtype=torch.float32):
        self.cfg = cfg
        self.sd = omodel.clone_state(state_dict, dtype)
        self.lr, self.momentum, self.wd, self.r = lr, momentum, weight_decay, r
        self.bufs = {}
        self.first = True
        self.scored = [k[:-len(".running_mean")] for k in self.sd
                       if k.endswith(".running_mean") and k[:-len(".running_mean")] not in ignore_prune]
        self.eic = {n: 0 for n in self.scored}

    def params(self):
        return {k: v for k, v in self.sd.items() if v.is_floating_point() and v.requires_grad}

    def step(self, x, labels, dropout_mask=None, update=True):
        for p in self.params().values():
            p.grad = None
        outs, loss, lowres = omodel.seg_forward(self.sd, x, self.cfg, labels, True, dropout_mask)
        loss.backward()
        for n in self.scored:   # dcfp_pruning.step (pruners/dcfp_pruner.py:15-20)
            w = self.sd[n + ".weight"]
            self.eic[n] = scoring.eic_step(w.detach().float().numpy(), w.grad.float().numpy(),
                                           self.eic[n], self.r)
        if update:              # torch.optim.SGD (optimizer.py:24-25)
            with torch.no_grad():
                for k, p in self.params().items():
                    g = p.grad
                    if self.wd != 0:
                        g = g.add(p, alpha=self.wd)
                    if self.first:
                        self.bufs[k] = g.clone()
                    else:
                        self.bufs[k].mul_(self.momentum).add_(g)
                    p.add_(self.bufs[k], alpha=-self.lr)
            self.first = False
        return float(loss.detach()), outs, lowres
