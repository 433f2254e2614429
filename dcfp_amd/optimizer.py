"""Optimizer construction and LR schedule — API of optimizer.py:12-79.

`build_optimizer` returns FusedSGD: the torch.optim.SGD(momentum, weight_decay) update of the
reference (optimizer.py:24-25) executed as ONE multi-tensor HIP launch per step over a
device-resident pointer table (the reference issues four foreach launches over ~470 tensors).
param_groups / lr / weight_decay keep torch.optim semantics, so adjust_learning_rate works
unchanged."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import SgdEntry, SGD_CHUNK, check
from .arena import ParamArena


def check_keywords_in_name(name, keywords=()):
    return any(k in name for k in keywords)


def set_weight_decay(model, skip_list=(), skip_keywords=()):
    has_decay, no_decay = [], []
    for name, param in model.named_parameters():
        if not param.requires_grad:
            continue
        if (name in skip_list) or check_keywords_in_name(name, skip_keywords):
            no_decay.append(param)
        else:
            has_decay.append(param)
    if len(no_decay) > 0:
        print("**** some para wo decay ****")
    return [{"params": has_decay}, {"params": no_decay, "weight_decay": 0.0}]


class FusedSGD(torch.optim.Optimizer):
    """SGD with momentum and weight decay (dampening 0, no nesterov):
       g' = g + wd*p ; buf = momentum*buf + g' (buf starts at 0, which makes the first step buf = g' exactly,
       torch's clone) ; p -= lr*buf.

    Parameters, gradients and momentum buffers live in three flat arenas (dcfp_amd/arena.py): addresses are
    stable, so the device pointer table of a param group is built and uploaded ONCE (`table_rebuilds`
    counts uploads) and a step is one kernel launch per non-empty group; `zero_grad()` launches nothing."""

    def __init__(self, params, lr=1e-3, momentum=0.0, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))
        self._tables = {}
        self._arena = None
        self.table_rebuilds = 0

    def _all_params(self):
        return [p for g in self.param_groups for p in g["params"] if p.requires_grad]

    def arena(self):
        """The arena holding this optimizer's parameters, created on first use once they are on the GPU
        (the reference builds the optimizer before `seg_model.to(device)`, train.py:212-219)."""
        params = self._all_params()
        if not params or not all(p.is_cuda and p.dtype == torch.float32 for p in params):
            return None
        if self._arena is None or not self._arena.covers(params):
            self._arena = ParamArena.of(params)
            self._tables = {}
            for i, p in enumerate(self._arena.params):      # momentum restored before the arena existed
                old = self.state[p].get("momentum_buffer") if p in self.state else None
                view = self._arena.momentum_view(i)
                if old is not None and old.data_ptr() != view.data_ptr():
                    view.copy_(old)
                self.state[p]["momentum_buffer"] = view
        return self._arena

    def _table(self, gi, group, params):
        key = tuple((p.data_ptr(), p.grad.data_ptr(), self.state[p]["momentum_buffer"].data_ptr()) for p in params) \
            + (group["weight_decay"],)
        cached = self._tables.get(gi)
        if cached is not None and cached[0] == key:
            return cached[1], cached[2], cached[3]
        self.table_rebuilds += 1
        entries = (SgdEntry * len(params))()
        chunk = 0
        for i, p in enumerate(params):
            e = entries[i]
            e.param, e.grad, e.momentum_buf = p.data_ptr(), p.grad.data_ptr(), self.state[p]["momentum_buffer"].data_ptr()
            e.n, e.first_chunk, e.weight_decay = p.numel(), chunk, group["weight_decay"]
            chunk += (p.numel() + SGD_CHUNK - 1) // SGD_CHUNK
        host = torch.frombuffer(bytearray(bytes(entries)), dtype=torch.uint8)
        dev = host.to(params[0].device)
        self._tables[gi] = (key, dev, len(params), chunk)
        return dev, len(params), chunk

    def zero_grad(self, set_to_none: bool = True):
        """Same contract as torch.optim.Optimizer.zero_grad.  With the arena: set_to_none detaches the gradient
        views (no launch; the next backward's first write overwrites), the in-place flavour (torch 1.10's
        default, train.py:256) is ONE fill of the flat gradient buffer."""
        ar = self.arena()
        if ar is not None:
            return ar.zero_grad(set_to_none)
        return super().zero_grad(set_to_none=set_to_none)

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._tables = {}
        self._arena = None          # momentum buffers were replaced: re-adopt them into the arena on next use
        from . import ops
        ops.WEIGHT_EPOCH[0] += 1

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _lib.lib()
        self.arena()
        for gi, group in enumerate(self.param_groups):
            params = [p for p in group["params"] if p.grad is not None]
            if not params:
                continue
            for p in params:
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous() \
                        or not p.grad.is_contiguous():
                    raise RuntimeError("FusedSGD: parameters and grads must be contiguous CUDA fp32")
                if "momentum_buffer" not in self.state[p]:     # parameter outside the arena
                    self.state[p]["momentum_buffer"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            table, n, chunks = self._table(gi, group, params)
            stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            check(L.dcfp_sgd_momentum_f32(C.c_void_p(table.data_ptr()), n, chunks, float(group["lr"]),
                                          float(group["momentum"]), 0, stream), "sgd_momentum")
        from . import ops
        ops.WEIGHT_EPOCH[0] += 1
        ops.refresh_wp()          # one launch: the permuted copies the conv kernels read, for the new weights
        return loss


def build_optimizer(config, model):
    skip_keywords = config.no_decay.split(",") if getattr(config, "no_decay", None) is not None else []
    parameters = set_weight_decay(model, [], skip_keywords)
    if config.optim == "sgd":
        return FusedSGD(parameters, momentum=config.momentum, lr=config.learning_rate,
                        weight_decay=config.weight_decay)
    if config.optim == "adamw":
        b1, b2 = map(float, config.betas.split(","))
        return torch.optim.AdamW(parameters, betas=(b1, b2), lr=config.learning_rate,
                                 weight_decay=config.weight_decay)
    return None


def lr_poly(base_lr, iter, max_iter, power):
    return base_lr * ((1 - float(iter) / max_iter) ** power)


def lr_warmup(base_lr, iter, warmup_iter=1500, warmup_ratio=1e-6):
    if iter >= warmup_iter:
        return base_lr
    return base_lr * (1 - (1 - float(iter) / warmup_iter) * (1 - warmup_ratio))


def adjust_learning_rate(optimizer, learning_rate, i_iter, max_iter, power, warmup):
    lr = lr_poly(learning_rate, i_iter, max_iter, power)
    if warmup > 0:
        lr = lr_warmup(lr, i_iter, warmup_iter=warmup)
    for param_group in optimizer.param_groups:
        param_group["lr"] = lr
    return lr
