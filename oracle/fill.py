"""Deterministic closed-form tensors shared by the golden generator and the tests
(no RNG, no weight files: SURVEY.md §8(c) 'what the import pins')."""
import math

import torch


def _wave(n, a, b, dtype=torch.float64):
    i = torch.arange(n, dtype=torch.float64)
    return torch.cos(a * i + b).to(dtype)


def _noise(n, a, b):
    """Closed-form pseudo-random values in [-sqrt(3), sqrt(3)) (unit variance, no periodic
    structure): frac(sin(a*i + b) * 43758.5453) evaluated in float64."""
    i = torch.arange(n, dtype=torch.float64)
    u = torch.sin((a * i + b) % 6.283185307179586) * 43758.5453123
    u = u - torch.floor(u)
    return (u - 0.5) * (2.0 * math.sqrt(3.0))


def closed_form_state(state_dict, gamma_amp=0.2, gamma_mid=1.0):
    """Fill every tensor of a state_dict (in its own order) from cos(a*i + b):
    conv weights ~ kaiming scale, BN gamma = mid + amp*cos, beta / biases = 0.1*cos,
    running_mean = 0.05*cos, running_var = 1 + 0.1*cos."""
    out = {}
    keys = list(state_dict.keys())
    for idx, name in enumerate(keys):
        t = state_dict[name]
        n = t.numel()
        b = 0.1 * idx
        if name.endswith("num_batches_tracked"):
            out[name] = torch.zeros_like(t)
            continue
        if name.endswith("running_mean"):
            v = 0.05 * _wave(n, 0.37, b)
        elif name.endswith("running_var"):
            v = 1.0 + 0.1 * _wave(n, 0.53, b)
        elif t.dim() == 4:
            fan_in = t.shape[1] * t.shape[2] * t.shape[3]
            v = math.sqrt(2.0 / fan_in) * _noise(n, 12.9898 + 1e-3 * idx, 78.233 * (idx + 1))
        elif name.endswith(".weight") and (name[:-len("weight")] + "running_mean") in state_dict:
            v = gamma_mid + gamma_amp * _wave(n, 0.7, b)
        else:
            v = 0.1 * _wave(n, 1.3, 0.5 * b)
        out[name] = v.reshape(t.shape).to(t.dtype)
    return out


def closed_form_input(N, H, W, dtype=torch.float32):
    n = N * 3 * H * W
    x = 0.7 * _wave(n, 0.0137, 0.3) + 0.9 * _noise(n, 4.1414, 0.77)
    return x.reshape(N, 3, H, W).to(dtype)


def closed_form_labels(N, H, W, num_classes=19, ignore=255):
    n = torch.arange(N).view(N, 1, 1)
    h = torch.arange(H).view(1, H, 1)
    w = torch.arange(W).view(1, 1, W)
    lab = ((h // 3) * 7 + (w // 5) * 13 + n * 5 + (h * w) % 3) % (num_classes + 4)
    lab = lab.long()
    lab[lab >= num_classes] = ignore
    return lab
