"""DCFP importance scoring and mask generation — API of pruners/dcfp_pruner.py:7-95.

`dcfp_pruning` (online, every training step): per scored BatchNorm channel
    flag = (grad * gamma > 0);  t = flag*|grad| + (!flag)*eic;  eic = eic*r + t*(1-r)
(dcfp_pruner.py:15-20).  The reference runs ~10 element-wise launches per BN layer (113
layers for DeepLabv3-R101); here all layers are updated by ONE HIP launch over a pointer
table, with the reference's operation order pinned so scores are bit-identical for
identical (gamma, grad) and exact zeros stay exact.  Under data parallelism the BN-gamma
gradients are all-reduced with the rest of the gradient buckets before step() runs
(train.py:265-268), which is what makes the score identical on every rank.

`DCFPPruner` (offline, CPU like the reference's prune.py): two-group global threshold on the
exported scores -> per-layer out-masks with a minimum keep count."""
import ctypes as C

import torch
import torch.nn as nn

from .. import _lib
from .._lib import EicEntry, check
from .channel_pruner import ChannelPruner


class dcfp_pruning():
    def __init__(self, model, r=0.99, **kwards):
        self.r = r
        self.state_dict = {"eic": {}}
        self._names = []
        for name, m in model.named_modules():
            if isinstance(m, (nn.BatchNorm2d, nn.SyncBatchNorm)) and name not in model.ignore_prune_layer:
                # the reference starts from the Python int 0 (dcfp_pruner.py:13); a zero vector
                # gives the same first update: 0*r + t*(1-r)
                self.state_dict["eic"][name] = 0
                self._names.append(name)
        self._key = None
        self._table = None

    def _build(self, mods):
        dev = mods[0].weight.device
        total = sum(m.weight.numel() for m in mods)
        old = self.state_dict["eic"]
        arena = torch.zeros(total, dtype=torch.float32, device=dev)
        entries = (EicEntry * len(mods))()
        off = 0
        for i, (name, m) in enumerate(zip(self._names, mods)):
            n = m.weight.numel()
            view = arena[off:off + n]
            if isinstance(old.get(name), torch.Tensor):
                view.copy_(old[name].to(dev))
            old[name] = view
            e = entries[i]
            e.gamma, e.grad, e.eic, e.n = m.weight.data_ptr(), m.weight.grad.data_ptr(), view.data_ptr(), n
            off += n
        self._arena = arena
        host = torch.frombuffer(bytearray(bytes(entries)), dtype=torch.uint8)
        self._table = host.to(dev)

    def step(self, model):
        mods_by_name = dict(model.named_modules())
        mods = [mods_by_name[n] for n in self._names]
        if not mods:
            return
        for n, m in zip(self._names, mods):
            if m.weight.grad is None:
                raise RuntimeError(f"dcfp_pruning.step: {n}.weight.grad is None (call after backward)")
            if not m.weight.is_cuda:
                raise RuntimeError("dcfp_pruning.step runs on the HIP kernel: model must be on cuda")
        key = tuple((m.weight.data_ptr(), m.weight.grad.data_ptr()) for m in mods)
        if key != self._key:
            self._build(mods)
            self._key = key
        r32 = float(self.r)
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(_lib.lib().dcfp_eic_update_f32(C.c_void_p(self._table.data_ptr()), len(mods), r32,
                                             float(1 - self.r), stream), "eic_update")

    def get_eic(self):
        return self.state_dict

    def export_eic(self, path):
        """score.pth = {'eic': {bn_name: FloatTensor[C]}} (dcfp_pruner.py:25-26)."""
        out = {"eic": {k: (v.detach().cpu().clone() if isinstance(v, torch.Tensor) else v)
                       for k, v in self.state_dict["eic"].items()}}
        torch.save(out, path)


def bn_group(bn_layer):
    """dcfp_pruner.py:36-37: backbone layers form group 0, head layers group 1."""
    return 0 if bn_layer.startswith("backbone") else 1


def compute_thresholds(eic, norm_conv_links, except_layers, channels, global_percent):
    """dcfp_pruner.py:43-66: per group, concatenate the scores of the non-excepted BN layers,
    sort ascending, threshold = sorted[int(size * global_percent)]."""
    sizes = [0, 0]
    for bn in norm_conv_links:
        if bn not in except_layers:
            sizes[bn_group(bn)] += channels[bn]
    pools = [torch.zeros(s) for s in sizes]
    index = [0, 0]
    for bn in norm_conv_links:
        if bn not in except_layers:
            g, n = bn_group(bn), channels[bn]
            pools[g][index[g]:index[g] + n] = eic[bn]
            index[g] += n
    thresh = [0, 0]
    for g in range(2):
        if pools[g].numel() > 0:
            sorted_bn, _ = torch.sort(pools[g])
            thresh[g] = sorted_bn[int(sizes[g] * global_percent)]
    return thresh


def compute_out_masks(eic, norm_conv_links, except_layers, channels, thresh, layer_keep):
    """dcfp_pruner.py:68-92: mask = score > thresh[group]; if fewer than
    max(1, int(C*layer_keep)) survive, force the top-k by descending sort."""
    masks = {}
    for bn, conv in norm_conv_links.items():
        n = channels[bn]
        if conv in except_layers:
            continue
        score = eic[bn]
        mask = score.gt(thresh[bn_group(bn)]).float()
        min_keep = int(n * layer_keep) if int(n * layer_keep) > 0 else 1
        if int(torch.sum(mask)) < min_keep:
            _, order = torch.sort(score, descending=True)
            mask[order[:min_keep]] = 1.0
        masks[conv] = mask
    return masks


class DCFPPruner(ChannelPruner):
    def __init__(self, global_percent=0.8, layer_keep=0.01, except_start_keys=["head.fc"],
                 score_file="", **kwards):
        super().__init__(except_start_keys=list(except_start_keys))
        self.layer_keep = layer_keep
        self.global_percent = global_percent
        self.eic = torch.load(score_file, map_location="cpu")["eic"]

    def get_bn_group(self, bn_layer):
        return bn_group(bn_layer)

    def get_para_score(self, bn_layer):
        return self.eic[bn_layer]

    def _channels(self):
        return {bn: self.name2module[bn].weight.data.shape[0] for bn in self.norm_conv_links}

    def get_thresh(self):
        return compute_thresholds(self.eic, self.norm_conv_links, self.except_layers,
                                  self._channels(), self.global_percent)

    def gen_channel_mask(self):
        thresh = self.get_thresh()
        masks = compute_out_masks(self.eic, self.norm_conv_links, self.except_layers,
                                  self._channels(), thresh, self.layer_keep)
        for conv, mask in masks.items():
            m = self.name2module[conv]
            m.out_mask = mask.reshape(m.out_mask.shape)
