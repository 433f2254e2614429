"""CPU restatement of the DCFP importance score and the SGD / LR rules around it.
numpy float32 with the reference's operation order, so it is bit-exact with what
torch-CPU computes.

Follows: pruners/dcfp_pruner.py:15-20 (EIC update; r from train.py:216),
optimizer.py:24-25 + torch.optim.SGD (momentum, weight decay), optimizer.py:60-79 (poly LR)."""
import numpy as np


def eic_step(gamma, grad, eic_prev, r=0.999):
    """flag = (g*w > 0); t = flag*|g| + (!flag)*eic_prev; eic = eic_prev*r + t*(1-r).
    All products/sums are separate float32 roundings; python scalars r and (1-r) are cast to
    float32 the way torch casts a wrapped number for a float32 tensor."""
    g = np.asarray(grad, dtype=np.float32)
    w = np.asarray(gamma, dtype=np.float32)
    prev = np.zeros_like(g) if (np.isscalar(eic_prev) and eic_prev == 0) else np.asarray(eic_prev, np.float32)
    flag = (g * w) > 0
    t = flag.astype(np.float32) * np.abs(g) + (~flag).astype(np.float32) * prev
    r32 = np.float32(r)
    omr32 = np.float32(1 - r)
    return (prev * r32 + t * omr32).astype(np.float32)


def sgd_step(p, g, buf, lr, momentum=0.9, weight_decay=5e-4, first=False):
    """torch.optim.SGD (dampening 0, no nesterov) in float64-free float32 arithmetic."""
    p = np.asarray(p, np.float32); g = np.asarray(g, np.float32)
    if weight_decay != 0:
        g = g + np.float32(weight_decay) * p
    buf = g.copy() if first else (np.asarray(buf, np.float32) * np.float32(momentum) + g)
    return (p - np.float32(lr) * buf).astype(np.float32), buf.astype(np.float32)


def lr_poly(base_lr, it, max_iter, power):
    return base_lr * ((1 - float(it) / max_iter) ** power)


def lr_warmup(base_lr, it, warmup_iter=1500, warmup_ratio=1e-6):
    if it >= warmup_iter:
        return base_lr
    return base_lr * (1 - (1 - float(it) / warmup_iter) * (1 - warmup_ratio))
