"""Deterministic synthetic score.pth content shared by the golden generator and the tests:
cos-shaped scores with exact zeros (gate never opened), ties, and all-zero layers that only
the layer_keep rule rescues."""
import torch


def synthetic_scores(model):
    eic = {}
    idx = 0
    for name, mod in model.named_modules():
        if isinstance(mod, torch.nn.BatchNorm2d) and name not in model.ignore_prune_layer:
            n = mod.weight.numel()
            i = torch.arange(n, dtype=torch.float64)
            s = (1e-3 * (1.0 + torch.cos(0.77 * i + 0.31 * idx)) * (1 + (i % 4))).float()
            s[(i % 3 == 0)] = 0.0
            s[(i % 11 == 5)] = 1e-3
            if idx % 9 == 4:
                s[:] = 0.0
            eic[name] = s
            idx += 1
    return eic
